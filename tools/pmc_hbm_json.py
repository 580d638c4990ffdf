#!/usr/bin/env python3
"""profiles/<tag>_pmc_hbm.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh: HBM-side bytes per
launch of every kernel, the sum over the coefficient kernels of one step (what bench.py reports as
roofline.traffic) and the hash of the kernel sources they were measured on (bench.py drops the figure when
the sources have changed since).

  python tools/pmc_hbm_json.py <tag> gpurun_out/prof_<tag>/fetch/fetch_results.db gpurun_out/prof_<tag>/write/write_results.db
"""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha256():
    h = hashlib.sha256()
    for f in ("sr_kernels.hip", "sr_device.hpp", "sr_kernels.hpp"):     # the kernels; the host API does not change their traffic
        h.update(open(os.path.join(ROOT, "spectrobot_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(distinct dispatch_id), sum(value) from counters_collection "
                     "where counter_name = ? group by kernel_name", (counter,)).fetchall()
    return {n.replace("void ", "").split("(")[0]: v * 1024.0 / max(k, 1) for n, k, v in rows}   # KiB -> bytes per launch


def main():
    tag, fetch_db, write_db = sys.argv[1:4]
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    kernels = {k: {"fetch_bytes": f.get(k, 0.0), "write_bytes": w.get(k, 0.0)} for k in sorted(set(f) | set(w))}
    coef = [k for k in kernels if any(s in k for s in ("sr_farfield_kernel<false", "near_wings_kernel<false>",
                                                        "near_zones_kernel<512, 1, false>", "sr_s2m_kernel<false>",
                                                        "sr_m2m_kernel", "sr_m2l_kernel<false>"))]
    step = [k for k in kernels if k in coef or "sr_prep_kernel" in k or "sr_limb_kernel" in k or "sr_limb_split_kernel" in k
            or "sr_los_columns" in k]
    out = {
        "profile_tag": tag,
        "kernel_sources_sha256": sources_sha256(),
        "units": "bytes per launch; rocprofv3 FETCH_SIZE / WRITE_SIZE are KiB (x1024), raw: the guide's gfx950 x2 applies to "
                 "16 B/lane streaming reads, these kernels read 64 B scalar and 8-16 B/lane vector loads (calibration of "
                 "round 1 on sr_radiance_kernel: 128 MB of 8 B/lane reads are reported as 117 MB)",
        "source": "tools/profile.sh %s (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes; counter passes serialise "
                  "the kernels)" % tag,
        "per_kernel": kernels,
        "coefficient_kernels": coef,
        "coefficient_kernels_hbm_bytes_per_step": sum(kernels[k]["fetch_bytes"] + kernels[k]["write_bytes"] for k in coef),
        "step_hbm_bytes_incl_prep_and_radiance": sum(kernels[k]["fetch_bytes"] + kernels[k]["write_bytes"] for k in step),
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
