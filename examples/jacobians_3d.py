#!/usr/bin/env python3
"""Round-3 features end to end on synthetic inputs (needs an MI355X):

  1. radiances + per-layer temperature Jacobian + per-level VMR Jacobian of a ray batch in ONE pass per ray
     (engine.limb_rays_jacobians -> sr_limb_rays_jacobians_dev)
  2. the same through a 3-D atmosphere: a coefficient row per LOS step with the state at the local SZA along
     the path (synthetic.limb_los_3d), Jacobians still per altitude layer
  3. a nadir / slant view over a Planck surface (synthetic.slant_los)

  python examples/jacobians_3d.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from spectrobot_amd import engine, synthetic as syn      # noqa: E402
import bench_configs as bc                                # noqa: E402


def main():
    import torch
    engine.set_device(0)
    n, nl = 40000, 40
    grid, L, atm, e_lev = bc.ch4_case(n, n, nl, config_id=3, w0=2950.0)
    ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
    vmr = np.full(nl, 0.0148)
    tz = atm["z"][0] + 40.0 + 90.0 * np.arange(6)
    dT = 0.05

    def coeffs_and_dT(a):
        co = ls.abscoeff_layers(a["temps"], a["press"], tvib=a["tvib"])
        ap = ls.abscoeff_layers(a["temps"] + dT, a["press"], tvib=a["tvib"])
        am = ls.abscoeff_layers(a["temps"] - dT, a["press"], tvib=a["tvib"])
        return co, ((ap[0] - am[0]) / (2 * dT), (ap[1] - am[1]) / (2 * dT))

    # 1. 1-D atmosphere at one solar zenith angle
    a1 = bc.sza_atmosphere(atm, 60.0)
    L1 = syn.limb_los(atm["z"], atm["nd"], [vmr], tz)
    los1 = engine.LimbLOS(L1["seg_off"], L1["seg_layer"], L1["pt_off"], L1["x"], L1["nd"], L1["vmr"], col_scale=[syn.CH4_ISO_RATIO])
    W = bc.layer_vmr_weights(atm["z"], L1["alt"])
    co, dco = coeffs_and_dT(a1)
    torch.cuda.synchronize()
    t0 = time.time()
    rad, jT, jV = engine.limb_rays_jacobians(co, los1, dcoeffs=dco, par_gas=np.zeros(nl, np.int32), par_w=W)
    torch.cuda.synchronize()
    print("1-D: %d rays x %d points: radiances + dI/dT_k + dI/dVMR_k (%d layers each) in %.2f ms; max |dI/dT| %.3e"
          % (len(tz), n, nl, (time.time() - t0) * 1e3, float(jT.abs().max())))

    # 2. 3-D: the state follows the local SZA along every ray
    L3 = syn.limb_los_3d(atm["z"], atm["nd"], [vmr], tz, 60.0, 30.0 * np.arange(len(tz)))
    los3 = engine.LimbLOS(L3["seg_off"], L3["seg_layer"], L3["pt_off"], L3["x"], L3["nd"], L3["vmr"], col_scale=[syn.CH4_ISO_RATIO])
    a3 = bc.step_atmosphere(atm, L3["seg_alt_layer"], L3["seg_mu"])
    co3, dco3 = coeffs_and_dT(a3)
    rad3, jT3, jV3 = engine.limb_rays_jacobians(co3, los3, dcoeffs=dco3, par_gas=np.zeros(nl, np.int32), par_w=W,
                                                seg_jac_row=L3["seg_alt_layer"], n_jac_rows=nl)
    torch.cuda.synchronize()
    d = float(((rad3 - rad).abs() / rad.abs().amax(dim=1, keepdim=True)).max())
    print("3-D: %d LOS steps instead of %d layers; radiances differ from the 1-D run by up to %.1f %% of a ray's maximum "
          "(cos SZA along the first ray: %.2f ... %.2f)" % (len(L3["seg_layer"]), nl, 100 * d, L3["seg_mu"][0],
                                                           L3["seg_mu"][L3["seg_off"][1] - 1]))

    # 3. nadir and slant views over a 160 K surface
    Ls = syn.slant_los(atm["z"], atm["nd"], [vmr], [0.0, 45.0, 70.0])
    loss = engine.LimbLOS(Ls["seg_off"], Ls["seg_layer"], Ls["pt_off"], Ls["x"], Ls["nd"], Ls["vmr"], col_scale=[syn.CH4_ISO_RATIO],
                          initial_temperature=160.0)
    rn = engine.limb_rays(co, loss, grid=grid)
    print("nadir / 45 / 70 deg views over a 160 K surface: mean radiance %s erg s-1 cm-2 sr-1 / cm-1"
          % " ".join("%.3e" % float(v) for v in rn.mean(dim=1)))


if __name__ == "__main__":
    main()
