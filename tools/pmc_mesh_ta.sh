#!/bin/bash
# tools/pmc_mesh_ta.sh c3|c5 [spp]: texture-addresser / L1 (TA, TCP, TD) counter passes of the mesh configs, single stream,
# each pass its own run -> gpurun_out/pmta_<cfg>_<n>/ and a per-kernel sum table gpurun_out/pmta_<cfg>.txt
# (no TD_* pass: rocprofv3 aborted with signal 6 on it on this pool and the run had to be killed)
CFG=${1:-c5}; SPP=${2:-32}
export TMPDIR=/tmp MSK_STREAMS=1
OUT=$PWD/gpurun_out
PASSES=(
 "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUSY_avr TA_BUSY_max"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"
 "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"
)
i=0
for P in "${PASSES[@]}"; do
  D=$OUT/pmta_${CFG}_$i; rm -rf $D
  rocprofv3 --output-format csv --kernel-trace --pmc $P -d $D -o x -- python3 tools/prof_mesh.py $CFG $SPP > $D.log 2>&1 || echo "pass $i ($P) failed"
  i=$((i+1))
done
python3 - $OUT $CFG <<'PY'
import csv, collections, glob, sys
out, cfg = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float)
for f in glob.glob(f"{out}/pmta_{cfg}_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("msk::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
f0 = glob.glob(f"{out}/pmta_{cfg}_0/**/*_kernel_trace.csv", recursive=True)
if f0:
    for r in csv.DictReader(open(f0[0])):
        dur[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("msk::", "")] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
with open(f"{out}/pmta_{cfg}.txt", "w") as fo:
    for k, v in sorted(tot.items()):
        if not k.startswith("k_"): continue
        fo.write(f"{k}  total_ns(pass0)={dur.get(k, 0):.0f}\n")
        for c, x in sorted(v.items()): fo.write(f"    {c:45s} {x:.6g}\n")
print(open(f"{out}/pmta_{cfg}.txt").read())
PY
