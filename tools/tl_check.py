import sys, os, time, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.util_data import make_fewshot_dataset
from multilingual_kws_amd.embedding import input_data, transfer_learning as tl
d = make_fewshot_dataset(tempfile.mkdtemp())
ms = input_data.standard_microspeech_model_settings(3)
for bs, ep, lr in ((16, 3, 1e-3), (64, 4, 1e-3), (32, 4, 1e-2)):
    t = time.time()
    name, model, det = tl.transfer_learn("target", d["train"], d["val"], d["unknown"], ep, 1, bs, lr, False, 0, ms, "synthetic", "dense_2",
                                         bg_datadir=d["bg_dir"], verbose=0, seed=11)
    h = model.history
    print(bs, ep, lr, "time %.2f" % (time.time() - t), name)
    print("  loss", [round(v, 3) for v in h["loss"]], "acc", [round(v, 3) for v in h["accuracy"]], "val_acc", h["val_accuracy"])
    tp, _ = tl.evaluate_files_single_target(d["val"], 2, model, ms); up, _ = tl.evaluate_files_single_target(d["unknown"], 2, model, ms)
    print("  target conf on val", tp.mean(), "on unknown", up.mean())
