#!/bin/bash
# Profiling passes of one round, run ON THE GPU BOX (through gpurun) from the repo root:
#   tools/profile_all.sh [spp]        -> gpurun_out/prof_{kt,kt1,fetch,write,sq,sq2}/
# kt / kt1 : rocprofv3 --kernel-trace --stats of bench.py (4-stream default, and MSK_STREAMS=1: per-kernel durations without overlap)
# fetch / write / sq / sq2 : --pmc passes (their own runs, no tracing domains besides the kernel trace) of tools/prof_run.py,
#                            single stream so that the counters of a dispatch are that dispatch's alone
# tools/summarize_profiles.py <tag> then turns them into profiles/<tag>_*.
SPP=${1:-512}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
rm -rf $OUT/prof_kt $OUT/prof_kt1 $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq $OUT/prof_sq2
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof_kt -o kt -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/prof_kt.json 2> $OUT/prof_kt.log
MSK_STREAMS=1 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof_kt1 -o kt1 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/prof_kt1.json 2> $OUT/prof_kt1.log
export MSK_STREAMS=1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch -o fetch -- python3 tools/prof_run.py $SPP > $OUT/prof_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write -o write -- python3 tools/prof_run.py $SPP > $OUT/prof_write.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $OUT/prof_sq -o sq -- python3 tools/prof_run.py $SPP > $OUT/prof_sq.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/prof_sq2 -o sq2 -- python3 tools/prof_run.py $SPP > $OUT/prof_sq2.log 2>&1
ls $OUT/prof_*
