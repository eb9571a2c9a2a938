#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/r6split
for order in 16 1004 1008; do
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  (cd /tmp && SPLIT_ORDER=$order timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/r6split/o${order}_$set -o p -- python $R/tools/split_pmc.py > /dev/null 2>&1)
done
done
python - <<'PY'
import csv, glob, collections, json
out={}
shapes=[(7304,12288,4096),(7304,22016,4096),(7304,4096,11008)]
for order in (16,1004,1008):
    vals=collections.defaultdict(list); durs=[]
    for setn in ("FETCH_SIZE","WRITE_SIZE","GRBM_GUI_ACTIVE"):
        for f in glob.glob("gpurun_out/r6split/o%d_%s/**/*counter_collection.csv"%(order,setn), recursive=True):
            rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Dispatch_Id"]))
            for r in rows:
                if "gemm_kernel" in r["Kernel_Name"] and r["Counter_Name"]==setn:
                    vals[setn].append(float(r["Counter_Value"]))
                    if setn=="GRBM_GUI_ACTIVE": durs.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for i,(M,N,K) in enumerate(shapes):
        sl=slice(3*i+1,3*i+3)      # launches 2 and 3 of each shape (the first is cold)
        f=sum(vals["FETCH_SIZE"][sl])/2; w=sum(vals["WRITE_SIZE"][sl])/2
        hbm=2*f*1024+w*1024; alg=4*(M*K+N*K+M*N)
        us=sum(durs[sl])/2 if durs else None
        out["order%d_%dx%dx%d"%(order,M,N,K)]={"hbm_bytes_per_launch":int(hbm),"algorithmic_bytes":alg,"overfetch_ratio":round(hbm/alg,2),"profiled_us":round(us,1) if us else None}
json.dump(out, open("gpurun_out/round6_split_gemm_pmc.json","w"), indent=1)
print(json.dumps(out, indent=1))
PY
