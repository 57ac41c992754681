#!/usr/bin/env python3
"""Summarise the two --pmc passes of tools/pmc_traffic.sh into per-kernel average FETCH_SIZE / WRITE_SIZE per launch.
rocprofv3 reports both in KiB-sized units of 1 KB (counter description: "kilobytes"); the raw values are kept and
the bytes derived from them are labelled with the correction applied (MI355X_MICROARCH.md §HBM: FETCH_SIZE tallies
128-byte requests at 64 B for wide coalesced reads -> x2; narrower accesses and WRITE_SIZE are uncalibrated)."""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0]
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return acc


def main():
    out, cfg = sys.argv[1], sys.argv[2]
    fe, wr = load(out + "/fetch", "FETCH_SIZE"), load(out + "/write", "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        nf, f = fe.get(k, [0, 0.0])
        nw, w = wr.get(k, [0, 0.0])
        kernels[k] = {"launches": nf or nw, "fetch_kb_per_launch": f / nf if nf else None,
                      "write_kb_per_launch": w / nw if nw else None}
    meta = {"config": cfg, "unit": "KB (rocprofv3 FETCH_SIZE / WRITE_SIZE, raw, per launch)"}
    pairs = {"c4": 1000000, "c2": 50000, "c4a": 1000000, "long": 5000, "mix": 500000, "amb": 402000}.get(cfg)
    sweeps = [v["launches"] for k, v in kernels.items() if "sweep_lane_kernel" in k or "sweep_wave_kernel" in k or "tile_sweep_kernel" in k]
    if pairs and sweeps:  # what bench.py's roofline.traffic needs: E-steps profiled, workload size, the command
        meta.update({"pairs_per_gpu": pairs, "estep_count": max(sweeps),
                     "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --config %s "
                                "--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-exchange-loopback  [tools/pmc_traffic.sh]" % cfg})
    if cfg == "amb":
        meta["command"] = ("rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- carmel --train-cascade -HJ -M 5 on the tagging cascade, corpus x 400 "
                           "[tools/amb_traffic.sh]")
    # which build of the kernels these counters belong to: bench.py reports them only for the same sources
    import hashlib
    import os
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "carmel_amd", "csrc")
    meta["kernels_hip_sha16"] = hashlib.sha256(b"".join(open(os.path.join(csrc, f), "rb").read() for f in
                                                        ("kernels.hip", "tile_sweep.hip", "sweep_math.hpp"))).hexdigest()[:16]
    if cfg == "c5":  # the sampler's sweeps: one forest_commit_kernel launch per parallel sweep
        commits = [v["launches"] for k, v in kernels.items() if "forest_commit_kernel" in k]
        meta.update({"forests": 100000, "sweep_count": max(commits) if commits else 0,
                     "forest_hip_sha16": hashlib.sha256(open(os.path.join(csrc, "forest.hip"), "rb").read()).hexdigest()[:16],
                     "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --config c5 --steps 3 "
                                "--warmup 1 --no-cpu-baseline --no-secondary --no-exchange-loopback  [tools/pmc_traffic.sh]"})
    if cfg == "crp":  # the lattice sampler's parallel sweeps (tools/gibbs_profile.sh): one gibbs_commit_kernel launch per sweep
        commits = [v["launches"] for k, v in kernels.items() if "gibbs_commit_kernel" in k]
        meta.update({"blocks": 100500, "sweep_count": max(commits) if commits else 0,
                     "gibbs_sha16": hashlib.sha256(b"".join(open(os.path.join(csrc, f), "rb").read() for f in
                                                             ("gibbs_lane.hip", "gibbs_exact.hip", "gibbs.hip"))).hexdigest()[:16],
                     "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- carmel --crp --crp-parallel -M 40 on the tagging "
                                "cascade x 100  [tools/gibbs_profile.sh]"})
    meta["kernels"] = kernels
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
