// mlp_nb64.hip -- the kernels of mlp_nb.hip on 64-row tiles (four 16-row blocks): the activation tile of a 256-wide net is
// 67.6 KB instead of 84.5 KB, so TWO 4-wave workgroups share a CU (two waves per SIMD with independent barriers).
#define OSRL_NB_RB 4
#define OSRL_NB_LAUNCH osrl_launch_fwd_nb64
#include "mlp_nb.hip"
