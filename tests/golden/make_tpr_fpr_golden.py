"""Generates tests/golden/tpr_fpr_golden.json by running the REFERENCE's own tpr_fpr.py (pure Python, importable in the build
container) on seeded detection / ground-truth lists.  Only inputs + outputs are committed."""
import contextlib, importlib.util, io, json, os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_tpr", "/root/reference/multilingual_kws/embedding/tpr_fpr.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.default_rng(77)
cases_gt, cases_tf = [], []
for rep in range(12):
    kws = ["alpha", "beta", "gamma"][: 1 + rep % 3]
    dur_ms = 120_000
    gt = sorted([(str(rng.choice(kws)), float(int(rng.integers(0, dur_ms)))) for _ in range(int(rng.integers(1, 25)))], key=lambda x: x[1])
    found = []
    for k, t in gt:                                     # detections near most occurrences + spurious ones
        if rng.uniform() < 0.7:
            found.append([k, int(t + rng.integers(-2500, 2500)), float(rng.uniform(0.5, 1))])
    for _ in range(int(rng.integers(0, 12))):
        found.append([str(rng.choice(kws)), int(rng.integers(0, dur_ms)), float(rng.uniform(0.5, 1))])
    if rep % 4 != 3:
        found.sort(key=lambda d: d[1])                 # (every fourth case stays unsorted: pins the early-exit scan)
    tol = [1500, 750, 300][rep % 3]
    with contextlib.redirect_stdout(io.StringIO()):
        out = ref.get_groundtruth(found, kws, gt, tol) if rep % 2 else ref.get_groundtruth(found, kws, gt)
    cases_gt.append(dict(found=found, targets=kws, groundtruth=gt, tol=tol if rep % 2 else None, out=out))
    kw = kws[0]
    gt_times = [t for k, t in gt if k == kw]
    if gt_times:
        fw = [[k, t] for k, t, _ in found]
        with contextlib.redirect_stdout(io.StringIO()):
            res = ref.tpr_fpr(kw, 0.5 + 0.04 * rep, fw, gt_times, dur_ms / 1000, tol, num_nontarget_words=None if rep % 2 else 40 + rep)
        cases_tf.append(dict(keyword=kw, thresh=0.5 + 0.04 * rep, found=fw, gt_times=gt_times, duration_s=dur_ms / 1000, tol=tol,
                             nontarget=None if rep % 2 else 40 + rep, out=res))
json.dump({"source": "multilingual_kws/embedding/tpr_fpr.py run on seeded inputs", "get_groundtruth": cases_gt, "tpr_fpr": cases_tf},
          open(os.path.join(HERE, "tpr_fpr_golden.json"), "w"))
print(len(cases_gt), len(cases_tf))
