#!/usr/bin/env python3
"""Copy the artefacts of tools/gpu_round_report.sh (gpurun_out/report/) into profiles/ under this
round's names and print the numbers the docs quote.  Usage: python tools/collect_profiles.py r01"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "gpurun_out", "report")
P = os.path.join(ROOT, "profiles")


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def main(tag):
    ks = open(os.path.join(R, "kernel_stats.txt")).read().rstrip().split("\n")
    b = last_json(os.path.join(R, "prof_bench.json"))
    hdr = (f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 16 --warmup 4 "
           f"--no-cpu-baseline --no-ttft   (MI355X; tools/gpu_round_report.sh)\n"
           f"# bench line of this run: {b['value']} tokens/s, {b['ms_per_step']} ms/step; roofline: "
           f"{json.dumps(b['roofline'])}\n"
           f"# decode_mfma_* rows include 2x32 graph-capture warm-up launches at the padded fill length (min_us).\n")
    body = "\n".join(line for line in ks if not line.startswith("{"))
    tail = ("\n# steady-state rows above (\"#\"): launches > 100 us only for decode_mfma_* (drops the capture "
            "warm-ups).\n# attn+merge per layer = decode_mfma_* + decode_merge_kernel steady averages; bench.py's "
            "HIP-event\n# average of the same pair (roofline.avg_launch_ms) additionally contains the inter-kernel gap "
            "of its eager pass.\n")
    open(os.path.join(P, f"{tag}_bench_kernel_stats.txt"), "w").write(hdr + body + tail)
    for src, dst in (("bench_serve.json", "serve_trace.json"), ("bench_serve_prefix.json", "serve_trace_prefix512.json"),
                     ("mllama.log", "mllama11b.txt"), ("gemv.log", "gemv.txt"), ("sampling.log", "sampling.txt"),
                     ("extend_attn.log", "extend_attn.txt"), ("extend_stamps.log", "extend_stamps_dma.txt"),
                     ("allreduce.log", "allreduce_rehearsal.txt"), ("parity_lines.txt", "parity_lines.txt"),
                     ("bench_replicas2_refused.log", "replicas2_refused.txt"),
                     ("prefill_kernel_stats.txt", "prefill_kernel_stats.txt")):
        p = os.path.join(R, src)
        if os.path.exists(p):
            text = "".join(line for line in open(p) if "amdgpu.ids" not in line)
            open(os.path.join(P, f"{tag}_{dst}"), "w").write(text)
    lines = {}
    for f in ("bench_decode", "bench_prefill", "bench_serve", "bench_serve_prefix", "bench_bs1", "bench_bs8",
              "bench_bs32", "bench_ctx128", "bench_ctx1024", "bench_ctx4096", "bench_fp8kv", "bench_70b_rank",
              "bench_tp2_rehearsal_gloo", "bench_tp2_rehearsal_direct", "bench_replicas2_rehearsal", "bench_librows_auto", "bench_tp4_70b_full_depth"):
        p = os.path.join(R, f + ".json")
        if os.path.exists(p):
            d = last_json(p)
            lines[f] = d
            r = d.get("roofline") or {}
            print(f"{f:20s} {d['metric']:28s} {d['value']:>10} {d['ms_per_step']:>9} ms/step  "
                  f"attn {r.get('avg_launch_ms')} ms {r.get('achieved')} GB/s  "
                  + " ".join(f"{k}={d[k]}" for k in ("ttft_ms", "tpot_ms", "itl_ms", "ttft_p50_ms", "ttft_p99_ms", "prefill_tokens_per_sec",
                                                    "total_tokens_per_sec", "duration_s", "step_frac_of_hbm_roofline")
                             if k in d)
                  + (f" cpu={d['cpu_baseline']['value']}" if "cpu_baseline" in d else ""))
    json.dump(lines, open(os.path.join(P, f"{tag}_bench_lines.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
