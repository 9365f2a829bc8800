"""Environment batching (replaces reference runners.py:7-50 + emulator_runner.py:18-33).

Two implementations behind the calls PAACLearner makes (update_environments / wait_updated /
get_shared_variables):

  * `Runners`       -- host BaseEnvironment plugins.  Same shared-variable convention as the reference
                       (states u8 [N,84,84,4], rewards f32 [N], episode_over f32 [N], actions f32 [N,A]);
                       envs are stepped in the calling process (workers=0) or in `workers` forked
                       processes over shared memory, auto-reset on terminal exactly like
                       emulator_runner.py:24-31.  The "go" token and the completion barrier are POSIX
                       semaphores (one per worker + one shared) instead of the reference's pickling
                       multiprocessing.Queue pairs (runners.py:14-15,44-50): the round trip of a step over 8
                       workers costs about 30 us instead of a millisecond or more.
  * device batching -- environments that have a device twin (EnvironmentCreator.device_env_spec) never
                       touch the host: one paac_synth_step launch per time step (see paac.DeviceRollout).
"""
import multiprocessing as mp
from multiprocessing.sharedctypes import RawArray

import numpy as np

_CTYPES = {np.dtype(np.float32): "f", np.dtype(np.uint8): "B", np.dtype(np.float64): "d"}


def step_emulators(emulators, variables):
    """emulator_runner.py:24-31 for a slice of environments."""
    states, rewards, overs, actions = variables
    for i, (emulator, action) in enumerate(zip(emulators, actions)):
        new_s, reward, episode_over = emulator.next(action)
        if episode_over:
            states[i] = emulator.get_initial_state()
        else:
            states[i] = new_s
        rewards[i] = reward
        overs[i] = episode_over


def step_emulators_raw(emulators, variables):
    """The same protocol for environments that hand out RAW screens (AtariEmulator.next_raw / initial_raw): slot 0 of
    raw[i] is the step's screen pair and counts[i] == 1; after a terminal the environment is reset and its slots hold
    the pairs that build the initial observation (counts[i] == number of slots).  The GPU turns them into
    observations (paac_preprocess_stack)."""
    raw, counts, rewards, overs, actions = variables
    for i, (emulator, action) in enumerate(zip(emulators, actions)):
        pair, reward, episode_over = emulator.next_raw(action)
        if episode_over:
            raw[i] = emulator.initial_raw()
            counts[i] = raw.shape[1]
        else:
            raw[i, 0] = pair
            counts[i] = 1
        rewards[i] = reward
        overs[i] = episode_over


class EmulatorRunner(mp.Process):
    step = staticmethod(step_emulators)

    def __init__(self, id, emulators, variables, queue, barrier, stop_flag=None):
        """queue: this worker's "go" semaphore; barrier: the completion semaphore all workers share; stop_flag: shared
        int, non-zero = leave at the next token (the reference sends None through the queue, emulator_runner.py:21-23)."""
        super(EmulatorRunner, self).__init__()
        self.id, self.emulators, self.variables, self.queue, self.barrier = id, emulators, variables, queue, barrier
        self.stop_flag = stop_flag
        self.daemon = True

    def run(self):
        while True:
            self.queue.acquire()
            if self.stop_flag is not None and self.stop_flag.value:
                break
            self.step(self.emulators, self.variables)
            self.barrier.release()


class RawEmulatorRunner(EmulatorRunner):
    step = staticmethod(step_emulators_raw)


class Runners(object):
    def __init__(self, EmulatorRunner, emulators, workers, variables):
        self.emulators = list(emulators)
        self.workers = int(workers)
        self.step = EmulatorRunner.step      # the runner class decides how a slice of environments is stepped
        self.variables = [self._get_shared(v) for v in variables]
        self.runners = []
        if self.workers > 0:
            if len(self.emulators) % self.workers != 0:
                raise ValueError("emulator_counts must be divisible by emulator_workers (runners.py:17-18)")
            self.queues = [mp.Semaphore(0) for _ in range(self.workers)]
            self.barrier = mp.Semaphore(0)
            self.stop_flag = mp.RawValue("i", 0)
            per = len(self.emulators) // self.workers
            for w in range(self.workers):
                sl = slice(w * per, (w + 1) * per)
                self.runners.append(EmulatorRunner(w, self.emulators[sl], [v[sl] for v in self.variables],
                                                   self.queues[w], self.barrier, self.stop_flag))

    @staticmethod
    def _get_shared(array):
        array = np.ascontiguousarray(array)
        raw = RawArray(_CTYPES[array.dtype], array.size)      # u8 stays 1 byte per cell (the reference used c_uint)
        shared = np.frombuffer(raw, dtype=array.dtype).reshape(array.shape)
        shared[...] = array
        return shared

    def start(self):
        for r in self.runners:
            r.start()

    def stop(self):
        if self.runners:
            self.stop_flag.value = 1
        for r in self.runners:
            self.queues[r.id].release()

    def get_shared_variables(self):
        return self.variables

    def update_environments(self):
        if self.workers > 0:
            for q in self.queues:
                q.release()
        else:
            self.step(self.emulators, self.variables)

    def wait_updated(self):
        if self.workers > 0:
            for _ in range(self.workers):
                self.barrier.acquire()
