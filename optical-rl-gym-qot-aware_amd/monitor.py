"""Batched counterpart of ``utils.evaluate_heuristic`` (``utils.py:124-162``) and a writer for the
``stable_baselines3.Monitor`` CSV layout the reference's result folders use
(``examples/phy_frag_rmsa/us-results/logs_1400_200/*.monitor.csv``: a ``#{"t_start": ..., "env_id": ...}`` JSON header
line, then ``r,l,t,<info keywords>`` with one row per episode), so that ``examples/visualize_loads*.ipynb`` can read
results produced on the GPU.
"""
from __future__ import annotations

import json
import time
from typing import Dict, List, Optional, Sequence

import numpy as np

RMSA_INFO_KEYWORDS = ("episode_service_blocking_rate", "service_blocking_rate", "episode_bit_rate_blocking_rate",
                      "bit_rate_blocking_rate")


def write_monitor_csv(path: str, rows: Sequence[Dict], env_id: str, info_keywords: Sequence[str],
                      t_start: Optional[float] = None):
    """``rows``: dicts with r, l, t and the info keywords, one per episode."""
    with open(path, "w") as f:
        f.write("#" + json.dumps({"t_start": time.time() if t_start is None else t_start, "env_id": env_id}) + "\n")
        f.write(",".join(("r", "l", "t") + tuple(info_keywords)) + "\n")
        for row in rows:
            f.write(",".join(repr(float(row[k])) if isinstance(row[k], (float, np.floating)) else str(row[k])
                             for k in ("r", "l", "t") + tuple(info_keywords)) + "\n")


def evaluate_heuristic_batched(env, policy: str, n_eval_episodes: int = 10, monitor_path: Optional[str] = None,
                               env_id: str = "RMSA-v0", info_keywords: Sequence[str] = RMSA_INFO_KEYWORDS,
                               chunk: int = 1000):
    """Run ``n_eval_episodes`` episodes of ``policy`` on every env of a :class:`BatchedRMSAEnv` with the reference
    loop's semantics: ``env.reset()`` (episode counters only) before every episode, step until ``done``.  Returns
    (episode_rewards [episodes, B], episode_lengths [episodes, B], per-episode info arrays); optionally writes one
    Monitor CSV with the rows of env 0, then env 1, ... per episode."""
    B = env.batch_size
    t0 = time.time()
    rewards, lengths = [], []
    infos: Dict[str, List[np.ndarray]] = {k: [] for k in info_keywords}
    rows = []
    for _ in range(n_eval_episodes):
        env.reset(only_episode_counters=True)
        ep_r = np.zeros(B)
        ep_l = np.zeros(B, np.int64)
        active = np.ones(B, bool)
        # the pending request is already counted, so an episode is episode_length - 1 steps; all envs of
        # a batch share the episode length and therefore finish together
        left = env.episode_length - 1
        while left > 0:
            n = min(left, chunk)
            out = env.run(policy, n, outputs=("reward", "done"))
            ep_r += out["reward"].sum(axis=0)
            ep_l += n
            left -= n
            active &= ~out["done"][-1].astype(bool)
        assert not active.any(), "episode did not finish on every env"
        c = env.counters()
        nxt = env.requests()["bit_rate"].astype(np.int64)
        # the Monitor logs the info of the episode's last step, built before the next request is generated
        proc, eproc = c["services_processed"] - 1, c["episode_services_processed"] - 1
        req, ereq = c["bit_rate_requested"] - nxt, c["episode_bit_rate_requested"] - nxt
        vals = {"service_blocking_rate": (proc - c["services_accepted"]) / proc,
                "episode_service_blocking_rate": (eproc - c["episode_services_accepted"]) / eproc,
                "bit_rate_blocking_rate": (req - c["bit_rate_provisioned"]) / req,
                "episode_bit_rate_blocking_rate": (ereq - c["episode_bit_rate_provisioned"]) / ereq}
        t = time.time() - t0
        rewards.append(ep_r.copy()); lengths.append(ep_l.copy())
        for k in info_keywords:
            infos[k].append(np.asarray(vals[k]))
        for i in range(B):
            row = {"r": float(ep_r[i]), "l": int(ep_l[i]), "t": round(t, 6)}
            row.update({k: float(vals[k][i]) for k in info_keywords})
            rows.append(row)
    if monitor_path is not None:
        write_monitor_csv(monitor_path, rows, env_id, info_keywords, t_start=t0)
    return np.stack(rewards), np.stack(lengths), {k: np.stack(v) for k, v in infos.items()}


# tests/test_rmsa_threads_us.py:66-69 (the Monitor the reference wraps PhyRMSA-v0 in)
PHY_INFO_KEYWORDS = ("episode_service_blocking_rate", "service_blocking_rate", "episode_bit_rate_blocking_rate",
                     "number_cuts_total", "rss_total_metric", "total_path_length", "num_moves", "num_defrag_cycle",
                     "avrage_gsnr", "average_mod_level", "average_path_index", "path_index", "physical_paths",
                     "num_moves_groom")


def evaluate_phy_heuristic_batched(env, policy: str, n_eval_episodes: int = 10, monitor_path: Optional[str] = None,
                                   env_id: str = "PhyRMSA-v0", info_keywords: Sequence[str] = PHY_INFO_KEYWORDS,
                                   chunk: int = 1000):
    """``evaluate_heuristic(env, heuristic, n_eval_episodes)`` (``utils.py:124-162``) for every env of a
    :class:`BatchedPhyRMSAEnv` with a device policy, and the Monitor CSV of the reference's experiment scripts
    (``tests/test_rmsa_threads_us.py:56-126``): one row per episode with the info dict of the episode's LAST step
    (``phy_rmsa_env.py:319-348``).  ``average_mod_level`` is the true mean (the reference's accumulator wraps at 256 under
    NumPy >= 2, SURVEY 8c caveat 2).  Returns (episode_rewards [episodes, B], episode_lengths, info arrays)."""
    B = env.batch_size
    t0 = time.time()
    rewards, lengths = [], []
    infos: Dict[str, List[np.ndarray]] = {k: [] for k in info_keywords}
    rows = []
    last_outs = ("accepted", "done", "number_cuts_total", "rss_total_metric", "defrag_counters")
    for _ in range(n_eval_episodes):
        env.reset(only_episode_counters=True)
        ep_r = np.zeros(B)
        left = env.episode_length - 1      # the pending request is already counted
        while left > 1:
            n = min(left - 1, chunk)
            ep_r += env.run(policy, n, outputs=("accepted",))["accepted"].sum(axis=0)
            left -= n
        last = env.run(policy, 1, outputs=last_outs)
        ep_r += last["accepted"][0]
        assert last["done"][0].all(), "episode did not finish on every env"
        c = env.counters()
        nxt = env.requests()["bit_rate"].astype(np.int64)
        proc, eproc = c["services_processed"] - 1, c["episode_services_processed"] - 1
        req, ereq = c["bit_rate_requested"] - nxt, c["episode_bit_rate_requested"] - nxt
        st = env.episode_stats()
        phys, chans = st["physical_services_accepted"], st["channels_accepted"]
        dc = last["defrag_counters"][0].astype(np.int64)
        vals = {"service_blocking_rate": (proc - c["services_accepted"]) / proc,
                "episode_service_blocking_rate": (eproc - c["episode_services_accepted"]) / eproc,
                "bit_rate_blocking_rate": (req - c["bit_rate_provisioned"]) / req,
                "episode_bit_rate_blocking_rate": (ereq - c["episode_bit_rate_provisioned"]) / ereq,
                "number_cuts_total": last["number_cuts_total"][0], "rss_total_metric": last["rss_total_metric"][0],
                "total_path_length": st["total_path_length"] / (phys + 1),
                "num_moves": dc[:, 0] / 2 + dc[:, 1], "num_moves_groom": dc[:, 1], "num_defrag_cycle": dc[:, 2],
                "avrage_gsnr": st["total_gsnr"] / (chans + 1),
                "average_mod_level": st["total_modulation_level"] / (chans + 1),
                "average_path_index": st["total_path_index"] / (phys + 1),
                "path_index": st["total_path_index"], "physical_paths": phys}
        t = time.time() - t0
        ep_l = np.full(B, env.episode_length - 1, np.int64)
        rewards.append(ep_r.copy()); lengths.append(ep_l)
        for k in info_keywords:
            infos[k].append(np.asarray(vals[k]))
        for i in range(B):
            row = {"r": float(ep_r[i]), "l": int(ep_l[i]), "t": round(t, 6)}
            row.update({k: (int(vals[k][i]) if np.issubdtype(np.asarray(vals[k]).dtype, np.integer) else float(vals[k][i]))
                        for k in info_keywords})
            rows.append(row)
    if monitor_path is not None:
        write_monitor_csv(monitor_path, rows, env_id, info_keywords, t_start=t0)
    return np.stack(rewards), np.stack(lengths), {k: np.stack(v) for k, v in infos.items()}
