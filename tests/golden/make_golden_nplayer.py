#!/usr/bin/env python3
"""Golden vectors for kitchens with 1, 3 and 4 players (runs ONLY in the build container).

The reference's Python MDP and encoder are N-player (envs/overcooked2_reimplement.py:221-234 per-viewer player slots,
:261-263 dish rule, envs/overcooked2_env.py:29 `num_players`, layouts `multiplayer_schelling` and `simple_single`).  This
script imports the reference in place through make_golden.import_reference() and records data-only fixtures:

    layout_params_nplayer.json            get_base_layout_params for every case below
    np_random_<case>.npz                  1000 uniform-random joint actions (np.random.default_rng(0)), horizon 200
    np_guided_<case>.npz                  600 steps of make_golden.py's own heuristic (cooks and delivers)

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_nplayer.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden as mg  # noqa: E402

# case name -> (reference layout, max_num_players)
CASES = {"schelling4": ("multiplayer_schelling", None), "schelling3": ("multiplayer_schelling", 3), "single": ("simple_single", None)}


def rollout(oenv, omdp, layout, num_players, actions, horizon=200):
    """make_golden.rollout with the reference's `num_players` argument (envs/overcooked2_env.py:295)."""
    env = oenv.SimplifiedOvercooked(layout, horizon=horizon, num_players=num_players)
    t0, p0, ts0 = mg.dump_state(env.state)
    rec = dict(init_tiles=t0, init_players=p0, init_timestep=ts0, init_obs=mg.wh_obs(env, env.state),
               actions=np.asarray(actions, dtype=np.int32))
    T, P = len(actions), env.mdp.num_players
    tiles, players, tsteps, rew_pp, rews, dones, obss = [], [], [], [], [], [], []
    for a in actions:
        state, per_player = env.mdp.get_state_transition(env.state, [int(x) for x in a])
        env.state = state
        done = state.timestep >= env.horizon
        rew_pp.append(list(per_player)); rews.append(sum(per_player)); dones.append(done)
        if done:
            env.n_reset()
        t, p, ts = mg.dump_state(env.state)
        tiles.append(t); players.append(p); tsteps.append(ts)
        obss.append(mg.wh_obs(env, env.state))
    rec.update(tiles=np.stack(tiles), players=np.stack(players), timestep=np.array(tsteps, np.int32),
               reward_per_player=np.array(rew_pp, np.int32).reshape(T, P), reward=np.array(rews, np.int32),
               done=np.array(dones, np.uint8), obs=np.stack(obss))
    return rec


def guided(oenv, layout, num_players, T, seed, eps=0.25, horizon=200):
    """The heuristic of make_golden.guided_actions against an env built with `num_players`."""
    orig = oenv.SimplifiedOvercooked

    def with_players(name, horizon=200, **kw):
        return orig(name, horizon=horizon, num_players=num_players, **kw)
    oenv.SimplifiedOvercooked = with_players
    try:
        return mg.guided_actions(oenv, None, layout, T, seed, eps=eps, horizon=horizon)
    finally:
        oenv.SimplifiedOvercooked = orig


def main():
    oenv, omdp, _ = mg.import_reference()
    params = {}
    for case, (layout, maxp) in CASES.items():
        d = oenv.get_base_layout_params(layout, 200, max_num_players=maxp)
        params[case] = {k: (list(map(int, v)) if isinstance(v, (list, tuple)) else int(v)) for k, v in d.items()}
        P = params[case]["num_players"]
        rng = np.random.default_rng(0)
        acts = np.stack([rng.integers(0, 6, size=P) for _ in range(1000)]).astype(np.int32)
        rec = rollout(oenv, omdp, layout, maxp, acts)
        assert rec["obs"].shape[1:] == (P, params[case]["width"], params[case]["height"], 5 * P + 10)
        mg.save(f"np_random_{case}.npz", **rec)
        g = guided(oenv, layout, maxp, 600, seed=1)
        rec = rollout(oenv, omdp, layout, maxp, g)
        print(f"    guided {case} (P={P}): total reward {int(rec['reward'].sum())}, deliveries {(rec['reward_per_player'] >= 20).sum()}")
        mg.save(f"np_guided_{case}.npz", **rec)
    with open(os.path.join(HERE, "layout_params_nplayer.json"), "w") as f:
        json.dump(params, f, indent=1, sort_keys=True)
    print("  wrote layout_params_nplayer.json")


if __name__ == "__main__":
    main()
