#!/bin/bash
# tools/pmc_quick.sh TAG [spp]: FETCH_SIZE and WRITE_SIZE passes of tools/prof_run.py (single stream) with the library MSK_GPU_LIB selects,
# summed per kernel -> gpurun_out/pmcq_TAG.txt (bytes per launch; read bytes = 2 * 1024 * FETCH_SIZE on gfx950, write = 1024 * WRITE_SIZE)
TAG=$1; SPP=${2:-128}
export TMPDIR=/tmp MSK_STREAMS=1
OUT=$PWD/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pq_$c; rocprofv3 --output-format csv --kernel-trace --pmc $c -d $OUT/pq_$c -o x -- python3 tools/prof_run.py $SPP > $OUT/pq_$c.log 2>&1
done
python3 - $OUT $TAG <<'PY'
import csv, collections, glob, sys
out, tag = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: [0.0, 0.0, 0])
for k, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    f = glob.glob(f"{out}/pq_{c}/**/*_counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("msk::", "")
        res[n][k] += float(r["Counter_Value"]) * (2048 if k == 0 else 1024)
        if k == 0: res[n][2] += 1
with open(f"{out}/pmcq_{tag}.txt", "w") as fo:
    for n, (rd, wr, cnt) in sorted(res.items()):
        line = f"{n:40s} launches {cnt:4d}  read {rd/1e9:8.3f} GB  write {wr/1e9:8.3f} GB  per launch {(rd+wr)/max(cnt,1)/1e6:9.2f} MB"
        print(line); fo.write(line + "\n")
PY
