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
    # Calibration on this chip (tools/fetch_calib.sh, profiles/r03_fetch_calibration.txt): FETCH_SIZE reports exactly 1/2
    # of every vector (global_load) read, whatever its width or stride, and the exact bytes of scalar loads; WRITE_SIZE
    # is exact.  The exact-mode kernels read their records by scalar loads; every other kernel through vector loads.
    scalar_kernels = ("sr_abscoeff_wings_kernel<", "sr_abscoeff_cores_kernel")
    for k, v in kernels.items():
        v["fetch_factor"] = 1.0 if any(s_ in k for s_ in scalar_kernels) else 2.0
        v["fetch_bytes_corrected"] = v["fetch_bytes"] * v["fetch_factor"]
    total = lambda names, key: sum(kernels[k][key] + kernels[k]["write_bytes"] for k in names)
    out = {
        "profile_tag": tag,
        "kernel_sources_sha256": sources_sha256(),
        "units": "bytes per launch; rocprofv3 FETCH_SIZE / WRITE_SIZE are KiB (x1024).  Corrected as calibrated on this chip "
                 "(profiles/r03_fetch_calibration.txt, tools/fetch_calib.sh): FETCH_SIZE reports exactly 1/2 of every vector "
                 "(global_load) read -- 8 B/lane, 16 B/lane and the per-lane record-field gathers of the row walks alike -- "
                 "and the exact bytes of scalar loads; WRITE_SIZE is exact.  fetch_bytes_corrected = 2 x fetch_bytes for the "
                 "kernels that read through vector loads (an upper bound where part of a kernel's reads are scalar: the "
                 "far-field polynomial coefficients of the near wings kernel, index-table look-ups), 1 x for the exact-mode "
                 "kernels whose records are scalar loads.  The *_raw totals are the uncorrected sums (round 2's convention).",
        "source": "tools/profile.sh %s (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes; counter passes serialise "
                  "the kernels)" % tag,
        "per_kernel": kernels,
        "coefficient_kernels": coef,
        "coefficient_kernels_hbm_bytes_per_step": total(coef, "fetch_bytes_corrected"),
        "coefficient_kernels_hbm_bytes_per_step_raw": total(coef, "fetch_bytes"),
        "step_hbm_bytes_incl_prep_and_radiance": total(step, "fetch_bytes_corrected"),
        "step_hbm_bytes_incl_prep_and_radiance_raw": total(step, "fetch_bytes"),
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
