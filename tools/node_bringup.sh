#!/bin/bash
# One command for the day an 8 x MI355X node appears: every multi-device step of this repo in the order
# in which a failure is cheapest to understand, each with the line it is expected to print and the condition under which
# the following steps are skipped.  Nothing here has run across two devices yet (no multi-GPU box was available to any
# round): the code and its tests are launch-ready; this script is the launch.
#
#   bash tools/node_bringup.sh               on the node: needs >= 8 visible devices for the full run (>= 2 runs what fits)
#   bash tools/node_bringup.sh --rehearsal   on ONE GPU: the same steps end to end with the ranks sharing the device
#                                            (gloo / same-device IPC).  Every line then says REHEARSAL and NOTHING it
#                                            prints is a scaling measurement - bench.py refuses to call it one, and so
#                                            does this script.  No scaling number is produced or estimated here.
#
# Mirrors: one process per rank with its device set before its first HIP call (server/server.py:252-265,
# distributed/parallel_state.py:176-187; sp_decode_ranges() and the range kernel's LDS limit are per device of the
# process since ABI 9), RCCL through torch.distributed (backend "nccl"), 127.0.0.1 rendezvous.
# Outputs: gpurun_out/bringup/<step>.{log,json}; a summary table at the end; exit code = number of failed steps.
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0          # dmabuf IPC: RCCL and the direct all-reduce need it on this image
REHEARSAL=0
ONLY=""                                      # --only 0,1 / --only 2,3,4,5: a subset of the steps (a 20-minute box runs the rehearsal in two calls)
while [ $# -gt 0 ]; do
  case "$1" in
    --rehearsal) REHEARSAL=1 ;;
    --only) ONLY=",$2,"; shift ;;
    *) echo "usage: $0 [--rehearsal] [--only STEP[,STEP...]]"; exit 2 ;;
  esac
  shift
done
want() { [ -z "$ONLY" ] || case "$ONLY" in *",$1,"*) return 0 ;; *) return 1 ;; esac; }
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
OUT=gpurun_out/bringup
mkdir -p $OUT
NDEV=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "devices visible: $NDEV; rehearsal: $REHEARSAL"
if [ "$NDEV" -lt 1 ]; then echo "no GPU: nothing to bring up"; exit 1; fi
if [ $REHEARSAL -eq 0 ] && [ "$NDEV" -lt 2 ]; then
  echo "one device and no --rehearsal: every step below needs one device per rank.  Re-run with --rehearsal to walk the"
  echo "steps on this GPU (labelled, not a measurement)."
  exit 1
fi
FAILED=0
SUMMARY=()
# run NAME SECONDS CMD...: run a step, keep its log, record pass / fail; a failed step does not stop the script unless the
# caller says so (the comment at each step names what a failure there makes pointless)
run() {
  local name=$1 secs=$2; shift 2
  echo "=== $name: $*"
  timeout -k 10 "$secs" "$@" > $OUT/$name.log 2>&1
  local rc=$?
  if [ $rc -eq 0 ]; then SUMMARY+=("$name: ok"); else SUMMARY+=("$name: FAILED (rc $rc, see $OUT/$name.log)"); FAILED=$((FAILED + 1)); fi
  tail -3 $OUT/$name.log | cut -c1-400
  return $rc
}
line() { grep -h '^{"metric"' $OUT/$1.log | tail -1 > $OUT/$1.json; cut -c1-300 $OUT/$1.json; }
RH=""
[ $REHEARSAL -eq 1 ] && RH="--rehearsal"
# ranks that can have a device of their own (rehearsal: the process guard of a shared box allows 6 on one card -> at most 4)
fits() { [ $REHEARSAL -eq 1 ] && [ "$1" -le 4 ] && return 0; [ $REHEARSAL -eq 0 ] && [ "$1" -le "$NDEV" ]; }

# ---- step 0: the single-device suite's attention + plan tests (is this box's one-device path what the rounds measured?)
if want 0; then
run 00_single_device_tests 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_plan_overflow.py -x -q -m gpu ||
  { echo "the one-device path is broken on this box: nothing below would mean anything"; printf '%s\n' "${SUMMARY[@]}"; exit $FAILED; }
fi

# ---- step 1: the tensor-parallel tests.  On the node the 21 one-device-per-rank cases (world 2 / 4 / 8: RCCL and the
# direct all-reduce; prefill, eager decode, graph decode with the collectives inside the graph; Llama-3-70B's 64 / 8 heads
# over 8 ranks) RUN instead of being skipped: expected "N passed, 0 skipped" with >= 8 devices.  Under --rehearsal they
# stay skipped and the shared-device twins of the same flows run (expected "... passed, 21 skipped").
# If this fails: stop - the bench lines below exercise the same collectives with less to look at.
if want 1; then
run 01_tensor_parallel_tests 1500 python -m pytest tests/test_gpu_tensor_parallel.py -x -q -m gpu -rs ||
  { echo "tensor-parallel tests failed: fix before benchmarking"; printf '%s\n' "${SUMMARY[@]}"; exit $FAILED; }
grep -h "passed\|skipped" $OUT/01_tensor_parallel_tests.log | tail -1
fi

# ---- step 2: the headline config as N independent replicas (config 2 does not shard: weak scaling, no data-path
# collective, the ranks meet at the timing barriers).  Expected per N: one JSON line, "n_gpus": N, "scaling": "weak",
# "ranks_seen": N, "devices_seen": N (each rank reports host / device / UUID), "rehearsal": false, "rccl_ranks": N (one
# RCCL all-reduce over all replicas: they can talk over xGMI), "rccl_note": null.
# The driver computes efficiency itself from the per-N values; this script prints the lines and nothing derived from them.
for N in 1 2 4 8; do
  want 2 || continue
  fits $N || { SUMMARY+=("02_replicas_$N: skipped (needs $N devices)"); continue; }
  EXTRA=""
  [ $REHEARSAL -eq 1 ] && EXTRA="--layers 4 --steps 4 --warmup 1 --no-cpu-baseline --no-ttft"
  run 02_replicas_$N 900 python bench.py --gpus $N $RH $EXTRA && line 02_replicas_$N
done

# ---- step 3: config 4, Llama-3-70B sharded TP = 8 (8 q heads + 1 kv head per rank), bs 128, RCCL all-reduce, eager and
# under HIP-graph replay with the collectives captured inside the graph.  Expected: one JSON line with config.tp 8,
# config.backend "RCCL (nccl)", config.all_reduce "nccl", config.graph true (else config.graph_note says why),
# "devices_seen": 8, "allreduce_us_per_call", "allreduces_per_step": 161, eager / graph ms per step, tokens/s of the group.
# If RCCL fails here but step 1 passed: look at "rccl_note" in the step-2 lines (communicator creation) first.
TP=8
VIA=rccl
[ $REHEARSAL -eq 1 ] && TP=2 && VIA=gloo_standin      # (RCCL refuses two ranks on one device: the rehearsal's collectives go over gloo)
if ! want 3; then
  SUMMARY+=("03..05: not selected")
elif fits $TP; then
  EXTRA=""
  [ $REHEARSAL -eq 1 ] && EXTRA="--layers 8 --steps 8 --warmup 2"
  run 03_tp${TP}_${VIA} 1200 python bench.py --mode tp --gpus $TP --tp $TP --model llama3-70b $RH $EXTRA && line 03_tp${TP}_${VIA}
  RCCL_OK=$?
  # ---- step 4: the same with the direct IPC all-reduce (csrc/allreduce.hip: peers' buffers mapped by IPC handle, system-scope
  # flags over xGMI - unproven across devices, hence opt-in).  Only meaningful once step 3 gave a line to compare with:
  # expected config.all_reduce "direct IPC kernel (SP_CUSTOM_ALLREDUCE=1)" and "allreduce_us_per_call" to set against step
  # 3's (bit-identical sums on every rank are what step 1's tests assert).  A hang here ends at
  # SP_CUSTOM_ALLREDUCE_TIMEOUT_S (collective, fatal, reported by every rank).
  # (Under --rehearsal "graph_ms_per_step" of this step is seconds, not milliseconds: two processes time-sliced on ONE device,
  # each replaying a graph whose all-reduce kernels spin on the other's flags, wait out whole time slices.  It says nothing
  # about a node, where every rank's kernels are resident on their own device; the eager figure is the one to read there too
  # until the graph one has been seen.)
  if [ $RCCL_OK -eq 0 ]; then
    SP_CUSTOM_ALLREDUCE=1 SP_CUSTOM_ALLREDUCE_TIMEOUT_S=20 run 04_tp${TP}_direct 1200 python bench.py --mode tp --gpus $TP --tp $TP --model llama3-70b $RH $EXTRA && line 04_tp${TP}_direct
    DIRECT_OK=$?
    # ---- step 5: ... and with the all-reduce fused into the following residual-add + RMSNorm (160 of a step's 161 collectives).
    # Expected "allreduce_plus_norm_us": {"two_launches", "fused_kernel", "fused_in_model": true}.
    # Skipped unless step 4 passed: it is the same transport with one more kernel on it.
    if [ $DIRECT_OK -eq 0 ]; then
      SP_CUSTOM_ALLREDUCE=1 SP_CUSTOM_ALLREDUCE_FUSE_NORM=1 SP_CUSTOM_ALLREDUCE_TIMEOUT_S=20 run 05_tp${TP}_direct_fused 1200 python bench.py --mode tp --gpus $TP --tp $TP --model llama3-70b $RH $EXTRA && line 05_tp${TP}_direct_fused
    else
      SUMMARY+=("05_tp${TP}_direct_fused: skipped (the direct all-reduce did not pass)")
    fi
  else
    SUMMARY+=("04_tp${TP}_direct: skipped (no RCCL line to compare with)" "05_tp${TP}_direct_fused: skipped")
  fi
else
  SUMMARY+=("03_tp${TP}_${VIA}: skipped (needs $TP devices)")
fi

echo
echo "---- bring-up summary ($([ $REHEARSAL -eq 1 ] && echo 'REHEARSAL on shared devices: code path only, NOT a measurement' || echo "$NDEV devices"))"
printf '%s\n' "${SUMMARY[@]}"
[ $REHEARSAL -eq 1 ] && echo "REHEARSAL: no figure above describes xGMI, RCCL across devices, or scaling."
exit $FAILED
