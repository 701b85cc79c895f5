import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),re.sub(r'\(anonymous namespace\)::|void ','',r['Kernel_Name'])[:30], r.get('Queue_Id','?')) for r in rows]
ks.sort()
ticks=[i for i,k in enumerate(ks) if k[2].startswith('step_tick') or k[2].startswith('step_begin')]
# a steady-state GRAPH replay from the middle of the timed region (the trace ends with bench.py's eager roofline /
# in-step probes and the Trainer-API extras)
m=len(ticks)//2
i0,i1=ticks[m],ticks[m+1]
t0=ks[i0][0]
print("step duration us", (ks[i1][0]-t0)/1e3, "n kernels", i1-i0)
for s,e,n,q in ks[i0:i1]:
    print(f"{(s-t0)/1e3:7.1f} {(e-t0)/1e3:7.1f} {(e-s)/1e3:6.1f} q{q} {n}")
