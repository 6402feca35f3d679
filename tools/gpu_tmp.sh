cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -s KILL 100 rocprofv3 --kernel-trace --output-format csv -d $O/nf_trace -- $R/tools/node_fetch_probe.bin > $O/nf_trace.log 2>&1 || { tail $O/nf_trace.log; exit 1; }
timeout -s KILL 100 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum --output-format csv -d $O/nf_tcp -- $R/tools/node_fetch_probe.bin > $O/nf_tcp.log 2>&1 || { tail $O/nf_tcp.log; exit 2; }
timeout -s KILL 100 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/nf_tcc -- $R/tools/node_fetch_probe.bin > $O/nf_tcc.log 2>&1 || { tail $O/nf_tcc.log; exit 3; }
cd $R
python3 - <<PY
import csv, glob, collections
dur=collections.defaultdict(list)
for f in glob.glob("$O/nf_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[(r["Kernel_Name"][:12])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
cnt=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("nf_tcp","nf_tcc"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[r["Kernel_Name"][:12]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for k in dur:
    ds=[d for _,d in sorted(dur[k])]
    for table in range(3):
        sl=ds[table*5:(table+1)*5]; t=min(sl)*1e-9
        line="%s table %d: %.1f us" % (k, table, t*1e6)
        for c,v in cnt[k].items():
            vals=[x for _,x in sorted(v)][table*5:(table+1)*5]
            line+="  %s %.3g (%.1f /us)" % (c, vals[0], vals[0]/t*1e-6)
        print(line)
PY
