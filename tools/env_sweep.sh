#!/bin/bash
# HIP runtime environment switches against the step time (bench.py, bf16 headline only): one line per setting.
# usage (on the GPU box): bash tools/env_sweep.sh > gpurun_out/env_sweep.txt
Q="--no-companion --no-cpu-baseline --no-configs --no-roofline --no-dp-form --no-pmc --blocks 3"
run() {
  local tag="$1"; shift
  local out
  out=$(env "$@" timeout 300 python bench.py $Q 2>/dev/null | tail -1)
  echo "$tag | $(echo "$out" | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["timing"]["block_ms_per_step"])
except Exception as e:
    print("FAILED", e)')"
}
run base A=1
run base2 A=1
run dev_kernarg1 HIP_FORCE_DEV_KERNARG=1
run dev_kernarg0 HIP_FORCE_DEV_KERNARG=0
run pkt_capture0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pkt_capture1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run opt_flush0 AMD_OPT_FLUSH=0
run opt_flush1 AMD_OPT_FLUSH=1
run graph_batch1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
run graph_batch64 DEBUG_HIP_GRAPH_BATCH_SIZE=64
run force_graph_queues DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run kernarg_copy0 DEBUG_HIP_KERNARG_COPY_OPT=0
run direct_dispatch0 AMD_DIRECT_DISPATCH=0
run hw_queues1 GPU_MAX_HW_QUEUES=1
run hw_queues8 GPU_MAX_HW_QUEUES=8
run base3 A=1
