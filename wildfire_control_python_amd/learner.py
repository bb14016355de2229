"""Batched value-based learners on top of BatchedForestFire.

The reference trains ONE environment with ~65 Keras calls per step (DQN.py:95-119,
156-185).  Here the same four algorithms - DQN (DQN.py), SARSA (DQN_SARSA.py),
Dueling DQN (DQN_DUEL.py) and Dueling SARSA (DQN_BOTH.py) - act in all environments
of a batch at once, entirely on the GPU: observations, actions, rewards and the
replay memory are torch tensors, the environment is stepped through the
asynchronous device entry points (the step kernel writes the observation itself).
This is glue around the hot path, not the path: plain torch ops.

Same ingredients as the reference:
  network    Flatten -> Dense(50, sigmoid) -> Dense(n_actions, linear)          DQN.py:209-233
  dueling    two 50-unit sigmoid streams, q = v + (a - mean(a))                  DQN_DUEL.py:18-48
  optimiser  Adam(lr=alpha, clipvalue=1), loss MSE                               DQN.py:226-229
  targets    prediction = target(s); prediction[a] = r (done) | r + gamma*max target(s')   DQN.py:166-176
             SARSA: r + gamma * target(s')[a']                                   DQN_SARSA.py:118-122
  policy     epsilon-greedy, eps = min + (max-min) * exp(-decay * episode)       DQN.py:188-203
  target net copied every `target_update` steps                                  DQN.py:107-110
"""
import math

import numpy as np

ALGORITHMS = ("DQN", "SARSA", "DDQN", "DSARSA")


def make_network(torch, width, height, depth, n_actions, dueling):
    nn = torch.nn
    n_in = width * height * depth

    class Plain(nn.Module):
        def __init__(self):
            super().__init__()
            self.hidden = nn.Linear(n_in, 50)
            self.out = nn.Linear(50, n_actions)

        def forward(self, x):
            return self.out(torch.sigmoid(self.hidden(x.flatten(1))))

    class Dueling(nn.Module):
        def __init__(self):
            super().__init__()
            self.h_adv, self.adv = nn.Linear(n_in, 50), nn.Linear(50, n_actions)
            self.h_val, self.val = nn.Linear(n_in, 50), nn.Linear(50, 1)

        def forward(self, x):
            x = x.flatten(1)
            adv = self.adv(torch.sigmoid(self.h_adv(x)))
            val = self.val(torch.sigmoid(self.h_val(x)))
            return val + (adv - adv.mean(dim=1, keepdim=True))      # DQN_DUEL.py:35-38

    return Dueling() if dueling else Plain()


def td_targets(torch, target_net, s, a, r, sp, ap, done, gamma, sarsa):
    """The reference's regression targets for a batch (DQN.py:161-180, DQN_SARSA.py:108-127):
    the target network's own prediction for s, with entry `a` replaced by the TD target."""
    with torch.no_grad():
        pred = target_net(s).clone()
        nxt = target_net(sp)
        boot = nxt.gather(1, ap[:, None])[:, 0] if sarsa else nxt.max(dim=1).values
        td = torch.where(done, r, r + gamma * boot)
        pred[torch.arange(len(a), device=a.device), a] = td
    return pred


class BatchedLearner:
    def __init__(self, env, algorithm="DQN", replay_capacity=None, batch_size=None, store_per_step=1024,
                 seed=0):
        import torch
        assert algorithm in ALGORITHMS
        self.torch, self.env, self.algorithm = torch, env, algorithm
        self.sarsa = algorithm in ("SARSA", "DSARSA")
        md = env.METADATA
        self.n_actions = md["n_actions"]
        self.gamma, self.alpha = md["gamma"], md["alpha"]
        self.max_eps, self.min_eps, self.decay = md["max_eps"], md["min_eps"], md["eps_decay_rate"]
        self.target_update = md["target_update"]
        self.batch_size = batch_size or md["batch_size"]
        self.capacity = replay_capacity or md["memory_size"]
        self.store_per_step = min(store_per_step, env.batch)
        self.dev = torch.device("cuda", env.device)
        torch.manual_seed(seed)
        W, H, B = env.width, env.height, env.batch
        self.model = make_network(torch, W, H, 3, self.n_actions, algorithm in ("DDQN", "DSARSA")).to(self.dev)
        self.target = make_network(torch, W, H, 3, self.n_actions, algorithm in ("DDQN", "DSARSA")).to(self.dev)
        self.target.load_state_dict(self.model.state_dict())
        self.opt = torch.optim.Adam(self.model.parameters(), lr=self.alpha)
        # device-resident interface buffers of the environment
        self.obs = torch.empty((B, W, H, 3), dtype=torch.int8, device=self.dev)
        self.next_obs = torch.empty_like(self.obs)
        self.reward = torch.empty((1, B), dtype=torch.float64, device=self.dev)
        self.done = torch.empty((1, B), dtype=torch.uint8, device=self.dev)
        # replay memory (ring buffer), states kept as int8
        C = self.capacity
        self.m_s = torch.empty((C, W, H, 3), dtype=torch.int8, device=self.dev)
        self.m_sp = torch.empty_like(self.m_s)
        self.m_a = torch.empty(C, dtype=torch.int64, device=self.dev)
        self.m_ap = torch.empty(C, dtype=torch.int64, device=self.dev)
        self.m_r = torch.empty(C, dtype=torch.float32, device=self.dev)
        self.m_d = torch.empty(C, dtype=torch.bool, device=self.dev)
        self.m_len, self.m_pos = 0, 0
        self.episodes = torch.zeros(B, dtype=torch.float32, device=self.dev)
        self.steps = 0
        self.logs = {"loss": [], "mean_reward": [], "finished_episodes": 0, "episode_reward_sum": 0.0}
        self._ep_return = torch.zeros(B, dtype=torch.float64, device=self.dev)

    # -------------------------------------------------------------- policy
    def epsilon(self):
        ep = float(self.episodes.mean().item())
        return self.min_eps + (self.max_eps - self.min_eps) * math.exp(-self.decay * ep)      # DQN.py:199-203

    def choose_actions(self, obs, eps=None):
        torch = self.torch
        eps = self.epsilon() if eps is None else eps
        with torch.no_grad():
            greedy = self.model(obs.float()).argmax(dim=1)
        explore = torch.rand(obs.shape[0], device=self.dev) <= eps                            # DQN.py:188-196
        rnd = torch.randint(0, self.n_actions, (obs.shape[0],), device=self.dev)
        return torch.where(explore, rnd, greedy)

    # -------------------------------------------------------------- memory
    def remember(self, s, a, r, sp, ap, d):
        n = s.shape[0]
        idx = (self.m_pos + self.torch.arange(n, device=self.dev)) % self.capacity
        self.m_s[idx], self.m_sp[idx], self.m_a[idx], self.m_ap[idx] = s, sp, a, ap
        self.m_r[idx], self.m_d[idx] = r.float(), d
        self.m_pos = (self.m_pos + n) % self.capacity
        self.m_len = min(self.capacity, self.m_len + n)

    def replay(self):
        torch = self.torch
        idx = torch.randint(0, self.m_len, (self.batch_size,), device=self.dev)
        s, sp = self.m_s[idx].float(), self.m_sp[idx].float()
        y = td_targets(torch, self.target, s, self.m_a[idx], self.m_r[idx], sp, self.m_ap[idx], self.m_d[idx],
                       self.gamma, self.sarsa)
        loss = torch.nn.functional.mse_loss(self.model(s), y)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_value_(self.model.parameters(), 1.0)                         # Adam(clipvalue=1)
        self.opt.step()
        return float(loss.item())

    # -------------------------------------------------------------- main loop
    def learn(self, n_steps, reset=True):
        """n_steps batched environment steps (every env acts in every step)."""
        torch, env = self.torch, self.env
        if reset:
            env.reset()
            env.get_state_device(self.obs)
        env.set_stream(torch.cuda.current_stream().cuda_stream)       # order the env after torch's work
        actions = self.choose_actions(self.obs)
        prev_done = torch.zeros(env.batch, dtype=torch.bool, device=self.dev)
        for _ in range(n_steps):
            a8 = actions.to(torch.int8)
            env.step_device(a8, 1, self.reward, self.done, d_obs=self.next_obs)
            r, d = self.reward[0], self.done[0] != 0
            next_actions = self.choose_actions(self.next_obs)          # a' (used by SARSA targets)
            k = self.store_per_step
            sel = torch.randperm(env.batch, device=self.dev)[:k]
            sel = sel[~prev_done[sel]]          # an auto-reset step (terminal obs -> fresh episode) is no transition
            self.remember(self.obs[sel], actions[sel], r[sel], self.next_obs[sel], next_actions[sel], d[sel])
            if self.m_len > self.batch_size:
                self.logs["loss"].append(self.replay())
            self.steps += 1
            if self.steps % self.target_update == 0:
                self.target.load_state_dict(self.model.state_dict())
            self._ep_return += r
            if bool(d.any()):
                self.logs["finished_episodes"] += int(d.sum().item())
                self.logs["episode_reward_sum"] += float(self._ep_return[d].sum().item())
                self._ep_return[d] = 0
                self.episodes += d.float()
            self.logs["mean_reward"].append(float(r.mean().item()))
            self.obs, self.next_obs = self.next_obs, self.obs
            actions, prev_done = next_actions, d
        torch.cuda.synchronize()
        return self.logs
