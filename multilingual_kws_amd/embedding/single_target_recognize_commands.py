"""Drop-in for multilingual_kws/embedding/single_target_recognize_commands.py: the single-target
sliding-average detector applied to streaming softmax outputs (reference :54-207).  Host-side, sequential,
O(windows) scalar work -- it stays on the CPU; the GPU work is the window loop that feeds it."""
import collections

import numpy as np


class RecognizeResult(object):
    """What the detector reports for the latest window."""

    def __init__(self):
        self.found_command = "_silence_"
        self.score = 0
        self.is_new_command = False


class SingleTargetRecognizeCommands(object):
    """Averages the target-class confidence over the last `average_window_duration_ms` of windows and fires
    when it crosses `detection_threshold`, at most once per `suppression_ms`; a sub-threshold average flips
    the state back to "_silence_" under the same suppression rule."""

    def __init__(self, labels, average_window_duration_ms, detection_threshold, suppression_ms, minimum_count, target_id):
        self._labels = labels
        self._target_id = target_id
        self._average_window_duration_ms = average_window_duration_ms
        self._detection_threshold = detection_threshold
        self._suppression_ms = suppression_ms
        self._minimum_count = minimum_count
        self._previous_results = collections.deque()
        self._label_count = len(labels)
        self._previous_top_label = "_silence_"
        self._previous_top_time = -np.inf

    def process_latest_result(self, latest_results, current_time_ms, recognize_element):
        latest_results = np.asarray(latest_results)
        if latest_results.shape[0] != self._label_count:
            raise ValueError("The results for recognition should contain {} elements, but there are {} produced".format(
                self._label_count, latest_results.shape[0]))
        if len(self._previous_results) != 0 and current_time_ms < self._previous_results[0][0]:
            raise ValueError("Results must be fed in increasing time order, but receive a timestamp of {}, which was "
                             "earlier than the previous one of {}".format(current_time_ms, self._previous_results[0][0]))
        self._previous_results.append([current_time_ms, latest_results])
        time_limit = current_time_ms - self._average_window_duration_ms
        while time_limit > self._previous_results[0][0]:
            self._previous_results.popleft()
        how_many = len(self._previous_results)
        sample_duration = current_time_ms - self._previous_results[0][0]
        if how_many < self._minimum_count or sample_duration < self._average_window_duration_ms / 4:
            recognize_element.found_command = self._previous_top_label
            recognize_element.score = 0.0
            recognize_element.is_new_command = False
            return
        # mean target confidence over the window, accumulated as the reference does (score / count, summed in order)
        # (in float64 like the reference's np.zeros accumulator: float32 inferences must not make this a float32 sum
        # under NumPy 2 promotion rules -- a score within 1e-7 of the threshold could flip a decision)
        score = 0.0
        for _, res in self._previous_results:
            score += float(res[self._target_id]) / how_many
        above = score > self._detection_threshold
        label = self._labels[self._target_id] if above else "_silence_"
        if self._previous_top_label == "_silence_" or self._previous_top_time == -np.inf:
            since = np.inf
        else:
            since = current_time_ms - self._previous_top_time
        fire = above and label != self._previous_top_label and since > self._suppression_ms
        release = score < self._detection_threshold and label == "_silence_" and since > self._suppression_ms
        if fire or release:
            self._previous_top_label = label
            self._previous_top_time = current_time_ms
            recognize_element.is_new_command = True
        else:
            recognize_element.is_new_command = False
        recognize_element.found_command = label
        recognize_element.score = score
