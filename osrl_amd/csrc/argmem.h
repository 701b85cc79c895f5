// argmem.h -- device-resident argument blocks for the launches of a captured step.
//
// The fused-MLP kernels take 1.2-2.4 KB descriptors (osrl_mlp_t + row map + activation pointers) BY VALUE.  The HIP
// runtime places kernel arguments either in device memory (HIP_FORCE_DEV_KERNARG=1, the default where the host can
// write device memory through a large BAR) or in host memory -- and then every WAVE of every launch fetches its part of
// the descriptor over PCIe: measured on one MI355X box (profiles/r3_kernarg_ab.txt) the CPQ step drops from 2150 to
// 1690 steps/s, the VAE's dW launch (which shares the link with a 768-workgroup forward) goes from 50 to 95 us.
// The driver's round-2 number (1700) is that case to 0.4 %.
//
// A step engine's launches are static (same pointers, same sizes every step), so their descriptors can live in HBM:
//   record  pass  (the warm-up run torch's graph capture needs anyway): every launch that goes through slot() copies
//                 its argument struct into a host staging buffer (identical blocks are stored once);
//   upload        one host->device copy of the staging buffer (outside the capture);
//   replay  pass  (the capture itself): slot() looks the struct up BY CONTENT and returns its device address; the
//                 launch site then starts the "_p" variant of the kernel, whose only kernel argument is that pointer
//                 and which reads the descriptor through the constant address space (scalar loads, same code as the
//                 by-value kernel behind one extra 8-byte load).  An unknown block (a launch the record pass did not
//                 make) is counted and launched by value -- correct, just not arena-resident.
// The context is thread-local and opt-in (osrl_args_begin / osrl_args_end); without it every launch is by value as
// before, so the stateless C ABI is unchanged for callers that do not use it.
#pragma once
#include <stdint.h>
#include <string.h>

namespace osrl_argmem {

enum { kOff = 0, kRecord = 1, kReplay = 2 };

struct Arena {
  char* host;      // staging buffer (host memory)
  const char* dev; // its device copy (valid in replay mode)
  int64_t cap, used;
  int32_t mode, n_blocks, n_hits, n_misses;
};

Arena* current();  // the calling thread's arena (nullptr: none); defined in optim.hip

// Each block: [int64 size][payload padded to 64 bytes]
inline const void* slot_bytes(const void* a, int64_t size) {
  Arena* ar = current();
  if (!ar || ar->mode == kOff) return nullptr;
  const int64_t padded = (size + 63) & ~int64_t(63);
  int64_t off = 0;
  while (off < ar->used) {  // a few dozen blocks per step: linear search, capture time only
    int64_t sz;
    memcpy(&sz, ar->host + off, sizeof sz);
    const int64_t pay = off + 64;
    if (sz == size && memcmp(ar->host + pay, a, (size_t)size) == 0) {
      if (ar->mode == kReplay) {
        ar->n_hits++;
        return ar->dev + pay;
      }
      return nullptr;  // record: already stored
    }
    off = pay + ((sz + 63) & ~int64_t(63));
  }
  if (ar->mode == kReplay) {
    ar->n_misses++;
    return nullptr;
  }
  if (ar->used + 64 + padded > ar->cap) {
    ar->n_misses++;
    return nullptr;
  }
  memset(ar->host + ar->used, 0, 64);
  memcpy(ar->host + ar->used, &size, sizeof size);
  memcpy(ar->host + ar->used + 64, a, (size_t)size);
  if (padded > size) memset(ar->host + ar->used + 64 + size, 0, (size_t)(padded - size));
  ar->used += 64 + padded;
  ar->n_blocks++;
  return nullptr;
}

// device address of an identical, already uploaded copy of `a`, or nullptr (launch by value)
template <class A>
inline const void* slot(const A& a) {
  return slot_bytes(&a, (int64_t)sizeof(A));
}

}  // namespace osrl_argmem

// descriptors behind a pointer are read through the constant address space: always scalar loads, pointers loaded from
// them are known-global (same code as the by-value kernel, which reads the kernarg segment the same way)
#define OSRL_CAS __attribute__((address_space(4)))
