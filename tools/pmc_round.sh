#!/bin/bash
# A round's PMC traffic records (FETCH_SIZE / WRITE_SIZE in separate passes) of the decode attention kernels on the final
# kernel sources: in the model under graph replay (headline, config 4's rank shape) and kernel alone (five workloads).
# Three gpurun calls (a call is limited to 20 minutes); results under gpurun_out/pmc_round/ and gpurun_out/pmc_NAME/: copy to
# profiles/rNN_bench_pmc_{headline,70b_rank}.{json,txt} and profiles/rNN_decode_attn_pmc_NAME.{json,txt}.
#   bash tools/pmc_round.sh model | alone1 | alone2
set -o pipefail
case "$1" in
model)
  bash tools/pmc_bench.sh "llama3-8b|bs256|ctxuniform|kvauto" || exit 1
  mkdir -p gpurun_out/pmc_round && cp gpurun_out/pmc_bench/summary.txt gpurun_out/pmc_round/bench_pmc_headline.txt && cp gpurun_out/pmc_bench/bench_pmc.json gpurun_out/pmc_round/bench_pmc_headline.json
  bash tools/pmc_bench.sh "llama3-70b-tp8-rank|bs128|ctxuniform|kvauto" --model llama3-70b-tp8-rank --bs 128 || exit 1
  cp gpurun_out/pmc_bench/summary.txt gpurun_out/pmc_round/bench_pmc_70b_rank.txt && cp gpurun_out/pmc_bench/bench_pmc.json gpurun_out/pmc_round/bench_pmc_70b_rank.json ;;
alone1)
  bash tools/pmc_decode.sh headline "llama3-8b|bs256|ctxuniform|kvauto" "--chunks 768 --interleave" || exit 1
  bash tools/pmc_decode.sh hkv1 "llama3-70b-tp8-rank|bs128|ctxuniform|kvauto" "--bs 128 --Hq 8 --Hkv 1 --chunks 768 --interleave" || exit 1
  bash tools/pmc_decode.sh fp8 "llama3-8b|bs256|ctxuniform|kvfp8_e5m2" "--kv fp8 --chunks 768 --interleave" || exit 1 ;;
alone2)
  bash tools/pmc_decode.sh ctx128 "llama3-8b|bs256|ctx128|kvauto" "--ctx 128 --chunks 128 --interleave" || exit 1
  bash tools/pmc_decode.sh bs32 "llama3-8b|bs32|ctx1024|kvauto" "--bs 32 --ctx 1024 --chunks 256 --interleave" || exit 1 ;;
*) echo "usage: $0 model|alone1|alone2"; exit 2 ;;
esac
