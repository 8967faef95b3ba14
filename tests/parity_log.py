"""Machine-readable parity report of the GPU suite.  Every parity helper records what it measured --
razor fraction, radius / rectangle / depth-order flips, fp32-arbiter uses, largest forward error on the strict pixels,
largest relative gradient error per tensor, re-drawn sweep seeds -- under the id of the running test; tests/conftest.py
writes the lot to gpurun_out/parity_report.json at the end of the session (the GPU-box copy is committed under profiles/)."""
import json
import os
from collections import OrderedDict

RECORDS = OrderedDict()


def _test_id() -> str:
    return os.environ.get("PYTEST_CURRENT_TEST", "unknown").split(" ")[0]


def record(**fields) -> None:
    """Merges `fields` into the running test's record; numeric fields that repeat keep their maximum, `n_*` counters add."""
    rec = RECORDS.setdefault(_test_id(), OrderedDict(checks=0))
    for k, v in fields.items():
        if isinstance(v, dict) and not all(isinstance(vv, (int, float)) and not isinstance(vv, bool) for vv in v.values()):
            rec[k] = v   # a structured note (lists, nested reports): stored as it is
        elif isinstance(v, dict):
            cur = rec.setdefault(k, {})
            for kk, vv in v.items():
                cur[kk] = max(cur.get(kk, 0.0), float(vv))
        elif k.startswith("n_"):
            rec[k] = rec.get(k, 0) + int(v)
        elif isinstance(v, (int, float)) and not isinstance(v, bool):
            rec[k] = max(rec.get(k, v), v)
        else:
            rec[k] = v


def dump(path: str) -> None:
    if not RECORDS:
        return
    tot = {"tests": len(RECORDS)}
    for key in ("n_radius_flips", "n_rectangle_flips", "n_depth_order_flips", "n_fp32_arbiter_uses", "n_redrawn_seeds", "n_forward_checks",
                "n_backward_checks", "n_row_criterion_fp32_oracle_uses"):
        tot[key] = sum(int(r.get(key, 0)) for r in RECORDS.values())
    tot["max_razor_fraction"] = max((r.get("razor_fraction", 0.0) for r in RECORDS.values()), default=0.0)
    tot["max_flip_tile_frac"] = max((r.get("flip_tile_frac", 0.0) for r in RECORDS.values()), default=0.0)
    tot["max_forward_err_strict"] = max((r.get("max_forward_err_strict", 0.0) for r in RECORDS.values()), default=0.0)
    worst = {}
    for r in RECORDS.values():
        for k, v in r.get("max_rel_grad_err", {}).items():
            worst[k] = max(worst.get(k, 0.0), v)
    tot["max_rel_grad_err"] = worst
    for key in ("max_rel_l2_grad_err", "max_row_bad_frac", "unmasked_vs_fp32_max", "unmasked_vs_fp32_l2", "unmasked_vs_fp32_rows_beyond_1e3"):
        w = {}
        for r in RECORDS.values():
            for k, v in r.get(key, {}).items():
                w[k] = max(w.get(k, 0.0), v)
        tot[key] = w
    tot["n_needles_relaxed"] = sum(int(r.get("n_needles_relaxed", 0)) for r in RECORDS.values())
    tot["n_visible_gaussians_checked"] = sum(int(r.get("n_visible_gaussians", 0)) for r in RECORDS.values())
    tot["n_unmasked_backward_checks"] = sum(int(r.get("n_unmasked_backward_checks", 0)) for r in RECORDS.values())
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"summary": tot, "tolerances": {"forward_abs": 1e-4, "grad_rel": 1e-3, "grad_rel_l2": 1e-4, "row_bad_frac": 5e-3, "row_bad_rows_abs": 3, "row_bad_frac_hard_cap_and_fp32_oracle_fallback": 1e-2,
                                                  "unmasked_l2": 4e-4, "unmasked_max": 6e-3, "means2d_ulps": 1.0, "conics_rel": 2.4e-7,
                                                  "depths_rel": 2.4e-7}, "tests": RECORDS}, f, indent=1)
