#!/usr/bin/env python
"""HBM traffic per launch of the decode mat-vec kinds from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected
separately, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; FETCH_SIZE is doubled: the gfx950 counter reports
64-byte units as 32):   pmc_traffic.py <fetch_dir> <write_dir> ["what the run was"] > profiles/rNN_pmc_traffic.json"""
import glob
import json
import os
import sqlite3
import sys

# qkv: the fused launch (wq|wk|wv + attention, k_qkv_attn) where it ran, else the plain wq|wk|wv mat-vec; its traffic also holds
# the attention's K/V read and the granule polling of the attention workgroups
KINDS = {"qkv": "%k_qkv_attn%<0,%", "qkv_unfused": "%k_mmvq_big<0, 3, 1%", "wo": "%k_mmvq_big<0, 1, 0%", "gate_up": "%k_mmvq_big<0, 2, 1%",
         "down": "%k_mmvq_big<0, 1, 2%", "lm_head": "%k_mmvq_big<0, 0, 1%"}  # qkv: k_qkv_attn or its WO form k_qkv_attn_wo (whichever ran)


def avg(d, like, counter):
    con = sqlite3.connect(sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))[-1])
    rows = list(con.execute("select counter_value from pmc_events where name like ? and counter_name = ?", (like, counter)))
    if not rows:
        return None, 0
    return sum(r[0] for r in rows) / len(rows), len(rows)


out = {}
for kind, like in KINDS.items():
    f, n = avg(sys.argv[1], like, "FETCH_SIZE")
    w, _ = avg(sys.argv[2], like, "WRITE_SIZE")
    if f is None:
        continue
    out[kind] = {"fetch_size_KiB_raw": round(f, 1), "write_size_KiB_raw": round(w or 0.0, 1), "dispatches": n,
                 "hbm_bytes_per_launch": int(round(f * 2 * 1024 + (w or 0.0) * 1024))}
out["method"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over a LLaMA-7B Q4_0 decode (GGML_HIP_GRAPH=0); "
                 "bytes = FETCH_SIZE x 2 (gfx950 correction) x 1 KiB + WRITE_SIZE x 1 KiB, per launch")
if len(sys.argv) > 3:  # what the profiled run was, e.g. the context range its dispatches ran at
    out["run"] = sys.argv[3]
if len(sys.argv) > 4:  # mean number of positions in the K/V cache over the profiled dispatches (bench.py takes traffic_over_algo at it)
    out["context_positions"] = float(sys.argv[4])
print(json.dumps(out, indent=1))
