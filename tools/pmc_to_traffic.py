#!/usr/bin/env python3
"""Folds the per-config PMC summaries of tools/gpu_r03_pmc.sh (gpurun_out/pmc_traffic_<cfg>.json: HBM-side bytes per
kernel dispatch, FETCH_SIZE corrected x2 as MI355X_MICROARCH.md prescribes) into profiles/traffic_latest.json under
"configs", keyed by bench.py's kernel slot names, and copies the raw summaries to profiles/<tag>_pmc_traffic_<cfg>.json.
usage: python tools/pmc_to_traffic.py <tag>"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
CFG = {"c2": "c2", "c3": "c3", "c4": "c4", "c5chunk": "c5"}


def slot_of(name, cfg):
    """bench.py slot of a rocprof kernel name: <T, L, CG, MODE, IO> -> pass0 / pass1 (MODE 0 / 2), Bluestein fwd_pass0 /
    conv_pass / inv_pass1."""
    if "conv_kernel" in name:
        return "conv_pass"
    targs = [t.strip() for t in name[name.index("<") + 1:name.index(">")].split(",")]
    mode = int(targs[3]) if len(targs) > 3 else -1
    if cfg == "c4":
        return {0: "fwd_pass0", 2: "inv_pass1"}.get(mode)
    return {0: "pass0", 2: "pass1"}.get(mode)


def main(tag):
    tpath = os.path.join(P, "traffic_latest.json")
    out = json.load(open(tpath)) if os.path.exists(tpath) else {}
    out.setdefault("configs", {})
    for cfg, key in CFG.items():
        src = os.path.join(G, f"pmc_traffic_{cfg}.json")
        if not os.path.exists(src):
            continue
        shutil.copy(src, os.path.join(P, f"{tag}_pmc_traffic_{cfg}.json"))
        rows = {}
        for name, r in json.load(open(src))["kernels"].items():
            s = slot_of(name, key)
            if s and "hbm_side_bytes" in r:
                rows[s] = {"rocprof_name": name, "hbm_side_bytes": r["hbm_side_bytes"], "read_bytes": r["read_bytes"],
                           "write_bytes": r["write_bytes"], "l2_hit_rate": r.get("l2_hit_rate")}
        out["configs"][key] = {"source": f"profiles/{tag}_pmc_traffic_{cfg}.json", "kernels": rows}
        if key == "c2":  # the default bench line's roofline.traffic reads per_launch_bytes[slot]
            out["note"] = ("HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes over "
                           "tools/run_config.py (gfx950: FETCH_SIZE tallies 128-B requests at 64 B); see configs.*.source")
            out["per_launch_bytes"] = {s: v["hbm_side_bytes"] for s, v in rows.items()}
            out.pop("raw_kb_per_dispatch", None)
        print(key, {k: round(v["hbm_side_bytes"] / 1e9, 2) for k, v in rows.items()})
    json.dump(out, open(tpath, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03")
