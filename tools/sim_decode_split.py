#!/usr/bin/env python3
"""Offline gate for a host-side choice of the decode split size by simulated makespan.

Model: a decode launch is a list of workgroups, one per (request, split, kv head group); the plan lists items longest
first and the hardware hands the next workgroup to the next free slot (CUs x waves-per-SIMD workgroups of 4 waves run at
once).  Two cost models:
  * "slot": a workgroup costs  t0 + keys * t_key  at an even share of the stream (valid only while every slot is busy);
  * "fluid" (the one the gate uses): HBM is one shared stream of BW bytes/us; a running workgroup first sits out t0 (its
    start chain index -> gather -> first tile, no bandwidth used), then drains its bytes at min(r_max, BW / active);
    split requests also write and re-read their partials (Hq/Hkv-group x (D + 1) x 4 B per workgroup) and the merge
    launch costs t_merge when anything was split.
The shipped rule (attention.py:_plan_chunk: power-of-two floor of sum * groups / 256, capped to [64, 768]) is compared
with the best candidate split size.

  python tools/sim_decode_split.py            # the headline batch and the other bench shapes, seed 0
No GPU involved.  The numbers this printed in round 5 are in profiles/NOTES.md.
(This models the (request, split) ITEM geometry.  Since the second half of round 5 a decode step of the default configuration runs
the range geometry - equal pieces of the step's keys, no split size to choose - and the items only serve soft-cap / fp32 / plan-less
launches: profiles/r05_decode_range.txt.)"""
import argparse
import heapq

import torch


def pow2_floor(x):
    return 1 << (max(int(x), 1).bit_length() - 1)


def shipped_chunk(total, groups, target=256, lo=64, hi=768):
    return max(lo, min(hi, pow2_floor(max(total, 1) * groups // target)))


def makespan(lens, chunk, wg_per_item, slots, t0, t_key):
    items = []
    for n in lens:
        full, tail = divmod(n, chunk)
        items += [chunk] * full + ([tail] if tail else [])
    items.sort(reverse=True)                       # the plan's order: longest first
    free = [0.0] * slots
    heapq.heapify(free)
    end = 0.0
    for keys in items:
        for _ in range(wg_per_item):
            t = heapq.heappop(free) + t0 + keys * t_key
            end = max(end, t)
            heapq.heappush(free, t)
    return end, len(items)


def fluid(lens, chunk, wg_per_item, slots, t0, bytes_per_key, bw, r_max, part_bytes, t_merge):
    """event-driven processor sharing: returns the launch's makespan in us"""
    items = []
    split = False
    for n in lens:
        full, tail = divmod(n, chunk)
        parts = [chunk] * full + ([tail] if tail else [])
        split = split or len(parts) > 1
        items += [(k, len(parts) > 1) for k in parts]
    items.sort(key=lambda kp: -kp[0])
    queue = [(k * bytes_per_key + (part_bytes if sp else 0)) for k, sp in items for _ in range(wg_per_item)]
    qi = 0
    now = 0.0
    waiting = []            # (ready time) heap of workgroups in their start chain: (t_ready, bytes)
    active = []             # remaining bytes of draining workgroups
    running = 0
    while qi < len(queue) or waiting or active:
        while running < slots and qi < len(queue):
            heapq.heappush(waiting, (now + t0, queue[qi]))
            qi += 1
            running += 1
        rate = min(r_max, bw / len(active)) if active else 0.0
        t_fin = now + min(active) / rate if active else float("inf")
        t_rdy = waiting[0][0] if waiting else float("inf")
        t_next = min(t_fin, t_rdy)
        dt = t_next - now
        if active:
            active = [r - rate * dt for r in active]
        now = t_next
        if t_rdy <= t_fin:
            active.append(heapq.heappop(waiting)[1])
        else:
            keep = [r for r in active if r > 1e-6]
            running -= len(active) - len(keep)
            active = keep
    return now + (t_merge if split else 0.0), len(items)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cus", type=int, default=256)
    ap.add_argument("--waves", type=int, default=3, help="workgroups resident per CU (launch bounds of the kernel)")
    a = ap.parse_args()
    slots = a.cus * a.waves
    shapes = [("headline bs 256 U[128,4096] Hkv 8", 256, (128, 4096), 8),
              ("bs 128 U[128,4096] Hkv 8", 128, (128, 4096), 8),
              ("bs 64 U[128,4096] Hkv 8", 64, (128, 4096), 8),
              ("70B rank bs 128 U[128,4096] Hkv 1", 128, (128, 4096), 1)]
    cands = [128, 192, 256, 320, 384, 448, 512, 576, 640, 704, 768, 896, 1024, 1280, 1536, 2048, 4096]
    for name, bs, (lo, hi), hkv in shapes:
        g = torch.Generator().manual_seed(0)
        lens = torch.randint(lo, hi + 1, (bs,), generator=g).tolist()
        total = sum(lens)
        wg = hkv // 4 if hkv % 4 == 0 else hkv     # workgroups per item (attention.py:_wg_groups)
        groups = 2 if hkv == 8 else 1              # attention.py:_head_groups at D = 128, 16-bit
        ship = shipped_chunk(total, groups)
        # t_key: the step streams at ~6.2 TB/s in total; a workgroup moves 4 (or 1) heads x 512 B per key
        bytes_per_key = (4 if hkv % 4 == 0 else 1) * 512
        for t0 in (4.0, 8.0):
            t_key = bytes_per_key / (6.2e6 / slots)                  # us per key at an even share of 6.2 TB/s
            res = {c: makespan(lens, c, wg, slots, t0, t_key) for c in sorted(set(cands + [ship]))}
            best = min(res, key=lambda c: res[c][0])
            ideal = total * wg * t_key / slots
            print(f"[slot ] {name}: sum {total}, t0 {t0:.0f} us: shipped chunk {ship} -> {res[ship][0]:7.1f} us ({res[ship][1]} items); "
                  f"best {best} -> {res[best][0]:7.1f} us ({100 * (res[ship][0] / res[best][0] - 1):+.1f} % for shipped); "
                  f"perfect balance {ideal:7.1f} us")
            print("    " + "  ".join(f"{c}:{res[c][0]:.0f}" for c in sorted(res)))
            for r_max in (16e3, 32e3):             # bytes/us a single workgroup can pull (one tile in flight per wave)
                part = (4 if hkv % 4 == 0 else 1) * (32 // 8 if hkv == 8 else 8) * 129 * 4 * 2   # written + re-read
                res = {c: fluid(lens, c, wg, slots, t0, bytes_per_key, 6.2e6, r_max, part, 7.0)
                       for c in sorted(set(cands + [ship]))}
                best = min(res, key=lambda c: res[c][0])
                print(f"[fluid] r_max {r_max / 1e3:.0f} GB/s: shipped {ship} -> {res[ship][0]:7.1f} us; best {best} -> "
                      f"{res[best][0]:7.1f} us ({100 * (res[ship][0] / res[best][0] - 1):+.1f} % for shipped)")
                print("    " + "  ".join(f"{c}:{res[c][0]:.0f}" for c in sorted(res)))


if __name__ == "__main__":
    main()
