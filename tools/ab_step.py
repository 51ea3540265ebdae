#!/usr/bin/env python3
"""Interleaved A/B of step-kernel builds on ONE device (cdna_hip_programming.md rule 24: deltas come from interleaved
rounds, never from separate boxes):  python tools/ab_step.py ROUNDS name1 name2 ...   (libhscn_<name>.so; "ship" =
libhscn.so).  Prints per variant the median / min ms per step of the default bench line and of the uniform-ids leg."""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "graph-hscn_amd", "graph_hscn", "lib")


def main():
    rounds = int(sys.argv[1])
    names = sys.argv[2:]
    extra = os.environ.get("AB_ARGS", "").split()
    res = {n: ([], [], []) for n in names}
    for r in range(rounds):
        for n in names:
            base, *envs = n.split(":")                      # "ship:HSCN_STEP_FIXED_LAYOUT=0": the same build under an env
            lib = os.path.join(LIB, "libhscn.so" if base == "ship" else f"libhscn_{base}.so")
            env = dict(os.environ, HSCN_LIB=lib)
            for kv in envs:
                k, v = kv.split("=", 1)
                env[k] = v
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "400", "--warmup", "40", "--no-cpu-baseline",
                                  "--no-streaming-spmm", "--no-stage-a-dense", "--repeats", "4"] + extra, env=env, stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, timeout=300)
            if out.returncode != 0:
                print(f"{n}: bench failed rc={out.returncode}", flush=True)
                continue
            d = json.loads(out.stdout.decode().strip().splitlines()[-1])
            res[n][0].append(d["repeats"]["median_ms_per_step_incl_first"])
            if d.get("other_cluster_ids"):
                res[n][1].append(d["other_cluster_ids"]["ms_per_step"])
            if d.get("stage_a"):
                res[n][2].append(d["stage_a"]["ms_per_step"])
            print(f"round {r} {n:36s} {1e3 * res[n][0][-1]:7.2f} us   uniform ids {1e3 * (res[n][1][-1] if res[n][1] else float('nan')):7.2f} us   stage A {1e3 * (res[n][2][-1] if res[n][2] else float('nan')):7.2f} us", flush=True)
    print("---- summary (us per step: median over rounds, min)")
    for n in names:
        a, b, c = res[n]
        if a:
            print(f"{n:36s} default {1e3 * statistics.median(a):7.2f} ({1e3 * min(a):7.2f})   uniform {1e3 * statistics.median(b):7.2f} ({1e3 * min(b):7.2f})"
                  + (f"   stage A {1e3 * statistics.median(c):7.2f} ({1e3 * min(c):7.2f})" if c else ""))


if __name__ == "__main__":
    main()
