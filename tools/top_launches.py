import json,sys
d=json.load(open("gpurun_out/bench_ops.json"))
calls=d["calls"]
tot=sum(c[2] for c in calls)
print("total ms", round(tot,3), "launches", len(calls))
calls.sort(key=lambda c:-c[2])
for c in calls[:int(sys.argv[1]) if len(sys.argv)>1 else 30]: print("%-22s %7.4f ms %6.1f TF/s  %s"%(c[0],c[2],c[1]/c[2]/1e9 if c[2] else 0,c[3]))
