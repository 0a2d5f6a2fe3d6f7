"""Developer tool: the clear Gaussians whose gradient error exceeds 1e-4 of the un-cancelled magnitude in
tests/test_gpu_fullsize_oracle.py::test_gradients_match_oracle_fullsize, with what is known about them.
   python tools/grad_offenders.py cfg3_1M_1080p"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from tests import test_gpu_fullsize_oracle as T
from tests.util import K7_KEYS, PARAM_KEYS, err_over_mass, k7_outputs, param_mass, splat_fields
name = sys.argv[1]
b = T.build_case(name, torch.device("cuda:0"))
pm, reach, clear = T.classify_gaussians(b)
N = b["N"]
ref7 = {k: np.asarray(b["bb"][k], np.float64).reshape(N, -1) for k in K7_KEYS}
got7 = k7_outputs(b["v_splats"])
pmass = param_mass(b["R"], b["Pn"], b["deg"], b["cb"], b["W"], b["H"], b["pc"]["radius"], b["m7"])
gotp = dict(zip(PARAM_KEYS, b["grads"]))
e7 = {k: err_over_mass(got7[k], ref7[k], b["m7"][k]) for k in K7_KEYS}
ep = {k: err_over_mass(gotp[k], b["pb"][k], pmass[k]) for k in PARAM_KEYS}
f = splat_fields(b["sp"], b["radii"])
tight = f["rect"].numpy().astype(np.int32); tight[(f["hits"] == 0).numpy()] = 0
pc = b["pc"]
mx, dropped = b["R"].dropped_pairs_max_alpha(pc["xy"], pc["conic"], pc["opac"], pc["rect"], pc["tiles_hit"], tight, b["cb"], b["W"], b["H"])
print(f"{name}: clear {clear.sum()}; Gaussians whose dropped tiles reach alpha >= 0.99/255: {(mx >= 0.99 / 255).sum()}")
bad = np.zeros(N, bool)
for k in K7_KEYS: bad |= clear & (e7[k] > 1e-4)
for k in PARAM_KEYS: bad |= clear & (ep[k] > 1e-4)
gmin, npix = b["R"].gaussian_min_margin(pc["xy"], pc["conic"], pc["opac"], pc["rect"], pc["tiles_hit"], b["cb"], b["W"], b["H"], pm, near=0.5)
for i in np.where(bad)[0]:
    print(f"g {i}: radius {pc['radius'][i]} (hip {int(f['radius'][i])}) rect {pc['rect'][i]} tight {tight[i]} xy {pc['xy'][i]} opac {pc['opac'][i]:.4f} conic {pc['conic'][i]} npix {npix[i]} gmin {gmin[i]:.2e} dropped max alpha*255 {mx[i] * 255:.3f}")
    print("   K7  " + " ".join(f"{k} {e7[k][i]:.1e}" for k in K7_KEYS))
    print("   par " + " ".join(f"{k} {ep[k][i]:.1e}" for k in PARAM_KEYS))
    print(f"   log_scales {b['Pn']['log_scales'][i]} v_ls ref {b['pb']['v_log_scales'][i]} hip {gotp['v_log_scales'][i]} mass {pmass['v_log_scales'][i]}")
    print(f"   v_conic ref {ref7['v_conic'][i]} hip {got7['v_conic'][i]} mass {b['m7']['v_conic'][i]}")
