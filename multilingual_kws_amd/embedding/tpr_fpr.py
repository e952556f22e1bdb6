"""Drop-in for multilingual_kws/embedding/tpr_fpr.py: matching of streaming detections against ground-truth times
(host-side bookkeeping behind run.py's detections dict and the streaming-accuracy summaries).

A detection at time t and a ground-truth occurrence at time g match when |t - g| <= time_tolerance_ms.  Both functions keep the
reference's results on its own inputs -- including two things that look like accidents and are pinned by vectors generated from the
reference (tests/golden/make_tpr_fpr_golden.py):
  * the reference scans SORTED lists and stops at the first entry past the window, so on an unsorted list an in-window entry behind
    an out-of-window one is not seen (`_in_window_sorted_scan`);
  * get_groundtruth (reference :1-59) returns from inside its loop over targets: only the FIRST target is ever classified.
"""


def _in_window_sorted_scan(times, centre, tol):
    """Whether some entry of `times` lies in [centre - tol, centre + tol] -- scanning in list order and giving up at the first
    entry above the window, as the reference's loops do (they assume ascending times)."""
    for t in times:
        if t > centre + tol:
            return False
        if t >= centre - tol:
            return True
    return False


def get_groundtruth(found_words, targets, groundtruth, time_tolerance_ms=1500, first_target_only=True):
    """found_words: [[keyword, time_ms, confidence], ...] (ascending times); groundtruth: [(keyword, time_ms), ...].
    -> list of dicts: misses as {keyword, time_ms, groundtruth: "fn"}, detections as {keyword, time_ms, confidence, groundtruth: "tp" | "fp"}.
    first_target_only=True is the reference as shipped (it returns after the first element of `targets`); False classifies every target."""
    out = []
    for target in targets:
        gt_times = [t for k, t in groundtruth if k == target]
        print("gt target occurences", len(gt_times))
        found = [f for f in found_words if f[0] == target]
        print("num found targets", len(found))
        found_times = [f[1] for f in found]
        out.extend(dict(keyword=target, time_ms=g, groundtruth="fn")
                   for g in gt_times if not _in_window_sorted_scan(found_times, g, time_tolerance_ms))
        out.extend(dict(keyword=target, time_ms=t, confidence=c,
                        groundtruth="tp" if _in_window_sorted_scan(gt_times, t, time_tolerance_ms) else "fp")
                   for _, t, c in found)
        if first_target_only:
            break
    return out


def tpr_fpr(keyword, thresh, found_words, gt_target_times_ms, duration_s, time_tolerance_ms, num_nontarget_words=None):
    """found_words: [[keyword, time_ms], ...]; gt_target_times_ms ascending.  -> the reference's summary dict (:62-135): true-positive
    rate, false accepts per hour, ... for one keyword at one threshold."""
    found_times = [t for w, t in found_words if w == keyword]
    n_gt = len(gt_target_times_ms)
    false_negatives = sum(not _in_window_sorted_scan(found_times, g, time_tolerance_ms) for g in gt_target_times_ms)
    true_positives = sum(_in_window_sorted_scan(gt_target_times_ms, t, time_tolerance_ms) for t in found_times)
    if true_positives > n_gt:          # several detections inside one occurrence's window (low thresholds)
        print("WARNING: weird timing issue")
        true_positives = n_gt
    false_positives = len(found_times) - true_positives
    result = dict(
        keyword=keyword,
        tpr=true_positives / n_gt,
        thresh=thresh,
        true_positives=true_positives,
        false_positives=false_positives,
        false_negatives=false_negatives,
        false_rejections_per_instance=false_negatives / n_gt,
        false_accepts_per_hour=false_positives / duration_s * 3600,
        groundtruth_positives=n_gt,
    )
    if num_nontarget_words is not None:
        result["fpr"] = false_positives / num_nontarget_words
    return result
