"""Generates tests/golden/detector_golden.json by running the REFERENCE's own detector
(/root/reference/multilingual_kws/embedding/single_target_recognize_commands.py -- pure numpy, importable
in the build container) on seeded score sequences.  Only the vectors (inputs + outputs) are committed."""
import importlib.util, json, os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location(
    "ref_strc", "/root/reference/multilingual_kws/embedding/single_target_recognize_commands.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

cases = []
rng = np.random.default_rng(2024)
configs = [dict(avg=100, thr=0.5, sup=500, minc=4, stride=20), dict(avg=500, thr=0.7, sup=300, minc=3, stride=20),
           dict(avg=100, thr=0.3, sup=500, minc=4, stride=40), dict(avg=60, thr=0.5, sup=0, minc=1, stride=20)]
for ci, cfg in enumerate(configs):
    for rep in range(3):
        n = 300
        # bursty target confidence: smooth random walk squashed to (0,1), plus unknown/silence split
        walk = np.cumsum(rng.standard_normal(n) * 0.6)
        tgt = 1 / (1 + np.exp(-(walk - walk.mean())))
        other = rng.uniform(0, 1, n) * (1 - tgt)
        probs = np.stack([1 - tgt - other, other, tgt], axis=1)
        rc = ref.SingleTargetRecognizeCommands(labels=["_silence_", "_unknown_", "kw"], average_window_duration_ms=cfg["avg"],
                                               detection_threshold=cfg["thr"], suppression_ms=cfg["sup"],
                                               minimum_count=cfg["minc"], target_id=2)
        el = ref.RecognizeResult()
        outs = []
        for i in range(n):
            rc.process_latest_result(probs[i], i * cfg["stride"], el)
            outs.append([el.found_command, float(el.score), bool(el.is_new_command)])
        cases.append({"config": cfg, "probs": probs.tolist(), "outputs": outs})
json.dump({"source": "multilingual_kws/embedding/single_target_recognize_commands.py:54-207 run on seeded inputs",
           "cases": cases}, open(os.path.join(HERE, "detector_golden.json"), "w"))
print(len(cases), sum(o[2] for c in cases for o in c["outputs"]), "new-command events")
