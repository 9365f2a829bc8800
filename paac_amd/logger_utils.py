"""Run bookkeeping (the role of reference logger_utils.py): args.json round trip and a JSONL metrics stream in
place of the TensorBoard summaries (paac.py:130-135,176-180; actor_learner.py:85-87).

metrics.jsonl, one object per line:
  {"kind": "progress", "global_step", "steps_per_s", "steps_per_s_avg", "last_10_rewards_avg", "lr", "grad_norm",
   "loss", "actor_loss", "critic_loss", "entropy", "time"}         -- every 2048/emulator_counts cycles (paac.py:172)
  {"kind": "gradients", "global_step", "global_norm", "raw_gradients": {"mean","stddev","max","min"},
   "clipped_gradients": {...}}                                     -- with every progress record: the summaries of
                                                                     actor_learner.py:85-87 / logger_utils.py:23-33
  {"kind": "episode", "global_step", "reward", "length"}          -- one per finished episode (paac.py:130-135)
"""
import json
import os
import time


def load_args(path):
    if path is None:
        return {}
    with open(path, 'r') as f:
        return json.load(f)


def save_args(args, folder, file_name='args.json'):
    d = {k: v for k, v in vars(args).items() if isinstance(v, (int, float, str, bool, type(None)))}
    if not os.path.exists(folder):
        os.makedirs(folder)
    with open(os.path.join(folder, file_name), 'w') as f:
        return json.dump(d, f)


class MetricsWriter(object):
    def __init__(self, folder, file_name='metrics.jsonl'):
        if not os.path.exists(folder):
            os.makedirs(folder)
        self.path = os.path.join(folder, file_name)
        self._f = open(self.path, 'a')

    def write(self, kind, **fields):
        rec = {"kind": kind, "time": time.time()}
        rec.update(fields)
        self._f.write(json.dumps(rec) + "\n")

    def flush(self):
        self._f.flush()

    def close(self):
        if not self._f.closed:
            self._f.close()
