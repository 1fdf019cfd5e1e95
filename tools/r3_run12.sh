#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
rm -rf /tmp/pl2
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/pl2 -- python3 tools/pmc_attn.py > /dev/null 2> $OUT/r3_pl2.err
echo rc=$?
python3 - <<'P'
import csv, glob, collections
rows=collections.OrderedDict()
for f in sorted(glob.glob("/tmp/pl2/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "attn" not in r["Kernel_Name"]: continue
        k=(int(r["Dispatch_Id"]), r["Kernel_Name"][r["Kernel_Name"].find("attn"):][:22], r["Grid_Size"])
        rows.setdefault(k,{})[r["Counter_Name"]]=rows.get(k,{}).get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
for k,v in rows.items():
    h,m=v.get("TCC_HIT_sum",0),v.get("TCC_MISS_sum",0)
    print(k[1],k[2],"L2 hit rate %.1f%%"%(100*h/max(h+m,1)),"req %.3e"%v.get("TCC_REQ_sum",0),"EA rdreq %.3e (x64B = %.1f MB)"%(v.get("TCC_EA0_RDREQ_sum",0), v.get("TCC_EA0_RDREQ_sum",0)*64/1e6))
P
