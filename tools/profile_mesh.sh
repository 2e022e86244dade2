#!/bin/bash
# Profiling passes of the mesh configs, run ON THE GPU BOX from the repo root:
#   tools/profile_mesh.sh c3|c5 [spp_trace] [spp_pmc]   -> gpurun_out/pm_<cfg>_{kt,kt1,fetch,write,sq,sq2,l2}/
# kt / kt1: rocprofv3 --kernel-trace --stats (default four loops; MSK_STREAMS=1), two renders each (the first allocates)
# fetch / write / sq / sq2 / l2: --pmc passes (their own runs), single stream, one render
CFG=${1:-c5}; SPPT=${2:-128}; SPPP=${3:-32}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
P=$OUT/pm_$CFG
rm -rf ${P}_kt ${P}_kt1 ${P}_fetch ${P}_write ${P}_sq ${P}_sq2 ${P}_l2
rocprofv3 --output-format csv --kernel-trace --stats -d ${P}_kt -o kt -- python3 tools/prof_mesh.py $CFG $SPPT 2 > ${P}_kt.log 2>&1 || exit 1
MSK_STREAMS=1 rocprofv3 --output-format csv --kernel-trace --stats -d ${P}_kt1 -o kt1 -- python3 tools/prof_mesh.py $CFG $SPPT 2 > ${P}_kt1.log 2>&1 || exit 1
export MSK_STREAMS=1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d ${P}_fetch -o fetch -- python3 tools/prof_mesh.py $CFG $SPPP > ${P}_fetch.log 2>&1 || exit 1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d ${P}_write -o write -- python3 tools/prof_mesh.py $CFG $SPPP > ${P}_write.log 2>&1 || exit 1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d ${P}_sq -o sq -- python3 tools/prof_mesh.py $CFG $SPPP > ${P}_sq.log 2>&1 || exit 1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM -d ${P}_sq2 -o sq2 -- python3 tools/prof_mesh.py $CFG $SPPP > ${P}_sq2.log 2>&1 || exit 1
rocprofv3 --output-format csv --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d ${P}_l2 -o l2 -- python3 tools/prof_mesh.py $CFG $SPPP > ${P}_l2.log 2>&1 || exit 1
tail -n 2 ${P}_*.log
