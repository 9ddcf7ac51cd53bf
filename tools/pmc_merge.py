#!/usr/bin/env python3
"""Merge the HBM-traffic figures of tools/pmc.sh runs (gpurun_out/pmc_<TAG>.json) into profiles/pmc_latest.json, each entry
stamped with the source hash of the build it was taken on (bench.py reports `roofline.traffic` only for entries of THE RUNNING
build).  Entries of other builds are dropped.

    tools/pmc_merge.py ROUND TAG:form:mode:model_err[:n_band[:evals_per_launch[:profile-file]]] ...
    e.g. tools/pmc_merge.py 5 "headline:k_hist<screen>:A:const" "general:k_hist<screen>:A:varying" "modeB:k_hist<exact>:B:const"
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frankenz_amd._lib import source_id

sid = source_id()
dst = os.path.join(ROOT, "profiles", "pmc_latest.json")
entries = []
for spec in sys.argv[2:]:
    f = spec.split(":")
    tag, form, mode, merr = f[:4]
    nband = int(f[4]) if len(f) > 4 and f[4] else 5
    epl = float(f[5]) if len(f) > 5 and f[5] else 1e11
    src = f[6] if len(f) > 6 else "gpurun_out/pmc_%s.txt" % tag
    d = json.load(open(os.path.join(ROOT, "gpurun_out", "pmc_%s.json" % tag)))
    if d.get("source_id") != sid:
        sys.exit("pmc_%s.json was taken on another build (%s, sources now %s)" % (tag, d.get("source_id"), sid))
    # the dominant kernel of the run: the largest traffic
    k, v = max(d["kernels"].items(), key=lambda kv: kv[1]["hbm_bytes_per_launch"])
    entries.append({"form": form, "mode": mode, "model_err": merr, "n_band": nband, "kernel": k, "evals_per_launch": epl,
                    "hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "fetch_KB": v["fetch_KB"], "write_KB": v["write_KB"],
                    "bytes_per_eval": v["hbm_bytes_per_launch"] / epl, "source_id": sid,
                    "source": "%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc.sh; %s); FETCH_SIZE x 2 "
                              "(gfx950, MI355X_MICROARCH.md) + WRITE_SIZE" % (src, d["bench_args"])})
json.dump({"round": int(sys.argv[1]), "source_id": sid, "entries": entries}, open(dst, "w"), indent=1)
print("wrote %d entries for build %s" % (len(entries), sid))
