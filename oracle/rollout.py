"""Oracle: the PAAC rollout/update cycle restated (TEST INFRASTRUCTURE, see __init__).

Follows:
  paac.py:59-183           PAACLearner.train: T lock-step env steps, batched policy, bootstrap,
                           float64 n-step return scan, t-major flatten, lr, one update per cycle
  paac.py:18-29            choose_next_actions: (v, pi) -> sample -> one-hot np.eye(A)[idx]
  emulator_runner.py:18-33 worker step: obs/reward/terminal; on terminal obs := get_initial_state()
  actor_learner.py:95-101  rescale_reward: clip to [-1, 1]
  actor_learner.py:119-123 get_lr: lr0 - step*lr0/anneal while step <= anneal else 0
PINNED by tests/golden/*.npz (captured from the reference's own train() loop).
"""
import numpy as np


def rescale_reward(r):
    if r > 1.0:
        return 1.0
    if r < -1.0:
        return -1.0
    return r


def get_lr(global_step, initial_lr, lr_annealing_steps):
    if global_step <= lr_annealing_steps:
        return initial_lr - (global_step * initial_lr / lr_annealing_steps)
    return 0.0


def nstep_returns(v_boot, rewards, masks, values, gamma):
    """paac.py:144-149.  All float64 like the reference's np.zeros buffers (paac.py:88-95);
    v_boot is the float32 network output promoted by the first multiply."""
    T = rewards.shape[0]
    y = np.zeros_like(rewards, dtype=np.float64)
    adv = np.zeros_like(rewards, dtype=np.float64)
    R = np.copy(v_boot)
    for t in reversed(range(T)):
        R = rewards[t] + gamma * R * masks[t]
        y[t] = np.copy(R)
        adv[t] = R - values[t]
    return y, adv


class OracleRollout:
    """One learner's rollout state.  `policy_fn(states_u8[N,84,84,4]) -> (v f32[N], pi f32[N,A])`,
    `sample_fn(pi) -> list of action indices` (consumes whatever RNG stream it closes over)."""

    def __init__(self, envs, num_actions, T, gamma, initial_lr, lr_annealing_steps,
                 policy_fn, sample_fn, global_step=0):
        self.envs = list(envs)
        self.N = len(self.envs)
        self.A = num_actions
        self.T = T
        self.gamma = gamma
        self.initial_lr = initial_lr
        self.lr_annealing_steps = lr_annealing_steps
        self.policy_fn = policy_fn
        self.sample_fn = sample_fn
        self.global_step = global_step
        self.shared_states = np.asarray([e.get_initial_state() for e in self.envs], dtype=np.uint8)  # paac.py:74
        self.shared_rewards = np.zeros(self.N, dtype=np.float32)
        self.shared_over = np.zeros(self.N, dtype=np.float32)
        self.total_episode_rewards = [0.0] * self.N
        self.emulator_steps = [0] * self.N
        self.finished_episodes = []   # (global_step, reward, length)

    def _step_envs(self, onehot):
        for i, env in enumerate(self.envs):                 # emulator_runner.py:24-31
            new_s, reward, over = env.next(onehot[i])
            if over:
                self.shared_states[i] = env.get_initial_state()
            else:
                self.shared_states[i] = new_s
            self.shared_rewards[i] = reward
            self.shared_over[i] = over

    def cycle(self):
        T, N, A = self.T, self.N, self.A
        rewards = np.zeros((T, N))
        states = np.zeros((T,) + self.shared_states.shape, dtype=np.uint8)
        actions = np.zeros((T, N, A))
        values = np.zeros((T, N))
        masks = np.zeros((T, N))
        pis = np.zeros((T, N, A), dtype=np.float32)
        for t in range(T):
            v, pi = self.policy_fn(self.shared_states)
            idx = self.sample_fn(pi)
            onehot = np.eye(A)[idx]
            actions[t] = onehot
            values[t] = v
            pis[t] = pi
            states[t] = self.shared_states
            self._step_envs(onehot)
            masks[t] = 1.0 - self.shared_over.astype(np.float32)
            for e in range(N):
                r = float(self.shared_rewards[e])
                self.total_episode_rewards[e] += r
                rewards[t, e] = rescale_reward(r)
                self.emulator_steps[e] += 1
                self.global_step += 1
                if self.shared_over[e]:
                    self.finished_episodes.append((self.global_step, self.total_episode_rewards[e], self.emulator_steps[e]))
                    self.total_episode_rewards[e] = 0.0
                    self.emulator_steps[e] = 0
        v_boot, _ = self.policy_fn(self.shared_states)
        y, adv = nstep_returns(v_boot, rewards, masks, values, self.gamma)
        lr = get_lr(self.global_step, self.initial_lr, self.lr_annealing_steps)
        return dict(states=states.reshape((T * N,) + self.shared_states.shape[1:]),
                    y=y.reshape(-1), adv=adv.reshape(-1), actions=actions.reshape(T * N, A),
                    lr=lr, values=values, rewards=rewards, masks=masks, pis=pis, v_boot=np.asarray(v_boot),
                    global_step=self.global_step)
