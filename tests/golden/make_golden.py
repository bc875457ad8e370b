#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, never on the GPU box).

Imports the reference's Python Overcooked MDP from /root/reference *in place*
(`envs/overcooked2_reimplement.py` -> `DummyMDP`, `envs/overcooked2_env.py` ->
`get_base_layout_params`, `SimplifiedOvercooked`; `pantheonrl_extension/vectorenv.py`
-> `SyncVectorEnv`), drives it with scripted, guided and uniform-random joint
actions and stores *data only* (inputs + expected outputs) as compressed .npz /
.json fixtures next to this script.  No reference source text is stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The fixtures pin the CPU oracle (`oracle/overcooked_oracle.c`) and, through it
and directly, the HIP path.  Serialisation used everywhere in this repo:

    tiles   int8 [S,4]  = (name, num_onions, num_tomatoes, cooking_tick)   name 0 = empty tile
    players int8 [P,6]  = (position, orientation, held.name, held.onions, held.tomatoes, held.tick)
    timestep int32
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True

LAYOUTS = ["simple", "unident_s", "random1", "random0", "random3", "mdp_test", "simple_tomato",
           "corridor", "scenario2_s"]
# this repo's own layout files (reference format): per-ingredient cook times / values, six pots, shaped
# rewards, a 3x4 grid and a 16x5 hall -- paths none of the reference's shipped layouts exercise
CUSTOM = ["six_pots", "tiny_mixed", "wide_hall"]
CLASSIC = ["simple", "unident_s", "random1", "random0", "random3"]


# ----------------------------------------------------------------------------- reference import
def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    """Register stand-ins for modules the reference imports but this path never uses
    (gym spaces, the Madrona python bindings, tensorboard), then import the real files."""
    class _Space:                      # only constructed, never used on this path
        def __init__(self, *a, **k):
            self.args = a
            self.n = a[0] if a else None
            self.shape = tuple(np.asarray(a[0]).reshape(-1)) if a else ()

    class _Env:
        pass

    spaces = _stub("gym.spaces", Space=_Space, MultiBinary=_Space, Discrete=_Space, Box=_Space)
    _stub("gym", Env=_Env, Space=_Space, spaces=spaces)
    _stub("build")
    _stub("build.madrona_python")
    _stub("build.madrona_simplecooked_example_python")
    import torch.utils  # noqa: F401
    _stub("torch.utils.tensorboard", SummaryWriter=object)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import envs.overcooked2_env as oenv
    import envs.overcooked2_reimplement as omdp
    import pantheonrl_extension.vectorenv as venv
    return oenv, omdp, venv


# ----------------------------------------------------------------------------- (de)serialisation
def obj_tuple(o):
    if o == 0:
        return (0, 0, 0, 0)
    return (o.name, o.num_onions, o.num_tomatoes, o._cooking_tick)


def dump_state(state):
    tiles = np.array([obj_tuple(o) for o in state.objects], dtype=np.int8)
    players = np.array([(p.position, p.orientation) + obj_tuple(p.held_object) for p in state.players],
                       dtype=np.int16)
    assert players.max() < 128
    return tiles, players.astype(np.int8), np.int32(state.timestep)


def load_state(omdp, tiles, players, timestep):
    def mk(t):
        if t[0] == 0:
            return 0
        o = omdp.ObjectState(int(t[0]), int(t[1]), int(t[2]))
        o._cooking_tick = int(t[3])
        return o
    ps = [omdp.PlayerState(int(p[0]), int(p[1]), mk(p[2:6])) for p in players]
    return omdp.OvercookedState(ps, [mk(t) for t in tiles], int(timestep))


def wh_obs(env, state):
    """(P, W, H, C) int8 exactly as SimplifiedOvercooked.get_obs hands it out."""
    obs = env.get_obs(state)
    out = np.stack([o[0] for o in obs])
    assert np.array_equal(out, out.astype(np.int8))
    return out.astype(np.int8)


# ----------------------------------------------------------------------------- rollout recorder
LAYOUT_PATHS = {}


def rollout(oenv, omdp, layout, actions, init=None, horizon=200):
    layout = LAYOUT_PATHS.get(layout, layout)
    """Steps the reference single env with the vector-wrapper reset rule
    (reference pantheonrl_extension/vectorenv.py:369-370): on done, n_reset() and hand out the
    reset observation while keeping the terminal reward / done."""
    env = oenv.SimplifiedOvercooked(layout, horizon=horizon)
    if init is not None:
        env.state = load_state(omdp, *init)
    t0, p0, ts0 = dump_state(env.state)
    rec = dict(init_tiles=t0, init_players=p0, init_timestep=ts0,
               init_obs=wh_obs(env, env.state), actions=np.asarray(actions, dtype=np.int32))
    T = len(actions)
    P = env.mdp.num_players
    tiles, players, tsteps, rew_pp, rews, dones, obss = [], [], [], [], [], [], []
    for a in actions:
        # per-player rewards (before the wrapper sums them) come from the MDP itself
        state, per_player = env.mdp.get_state_transition(env.state, [int(x) for x in a])
        env.state = state
        done = state.timestep >= env.horizon
        rew_pp.append(list(per_player))
        rews.append(sum(per_player))
        dones.append(done)
        if done:
            env.n_reset()
        t, p, ts = dump_state(env.state)
        tiles.append(t); players.append(p); tsteps.append(ts)
        obss.append(wh_obs(env, env.state))
    rec.update(tiles=np.stack(tiles), players=np.stack(players), timestep=np.array(tsteps, np.int32),
               reward_per_player=np.array(rew_pp, np.int32).reshape(T, P),
               reward=np.array(rews, np.int32), done=np.array(dones, np.uint8), obs=np.stack(obss))
    return rec


# ----------------------------------------------------------------------------- guided policy
DIRS = {0: lambda p, w: p - w, 1: lambda p, w: p + w, 2: lambda p, w: p + 1, 3: lambda p, w: p - 1}


def guided_actions(oenv, omdp, layout, T, seed, eps=0.25, horizon=200):
    """Own heuristic (not reference code): fetch onions -> pot, dish -> ready pot, soup -> serve,
    with eps-uniform noise so counters / collisions / useless interacts are exercised too.
    Produces the action stream; the reference MDP is the simulator it plans against."""
    rng = np.random.default_rng(seed)
    env = oenv.SimplifiedOvercooked(LAYOUT_PATHS.get(layout, layout), horizon=horizon)
    mdp = env.mdp
    W, S = mdp.width, mdp.size
    terr = list(mdp.terrain_mtx)
    P = mdp.num_players
    out = []

    def goals(player, state):
        held = player.held_object
        pots = [i for i in range(S) if terr[i] == 1]
        want = []
        if held == 0:
            busy = [i for i in pots if state.objects[i] != 0 and state.objects[i]._cooking_tick >= 0]
            if busy and rng.random() < 0.6:
                want = [i for i in range(S) if terr[i] == 4]
            else:
                want = [i for i in range(S) if terr[i] in (3, 6)]
        elif held.name in (1, 2):
            want = [i for i in pots if state.objects[i] == 0 or
                    (state.objects[i]._cooking_tick < 0 and
                     state.objects[i].num_onions + state.objects[i].num_tomatoes < 3)] or \
                   [i for i in range(S) if terr[i] == 2]
        elif held.name == 3:
            want = [i for i in pots if state.objects[i] != 0] or [i for i in range(S) if terr[i] == 2]
        else:
            want = [i for i in range(S) if terr[i] == 5]
        return set(want)

    def plan(player, state, others):
        want = goals(player, state)
        if not want:
            return int(rng.integers(0, 6))
        pos = player.position
        # facing a wanted tile -> interact
        if DIRS[player.orientation](pos, W) in want:
            return 5
        for d in range(4):
            if DIRS[d](pos, W) in want:
                return d          # turn towards it (blocked move = pure turn)
        # BFS over AIR tiles to any tile adjacent to a wanted tile
        prev = {pos: None}
        queue = [pos]
        hit = None
        while queue and hit is None:
            cur = queue.pop(0)
            for d in range(4):
                nxt = DIRS[d](cur, W)
                if 0 <= nxt < S and terr[nxt] == 0 and nxt not in prev and nxt not in others:
                    prev[nxt] = (cur, d)
                    if any(DIRS[e](nxt, W) in want for e in range(4)):
                        hit = nxt
                        break
                    queue.append(nxt)
        if hit is None:
            return int(rng.integers(0, 6))
        while prev[hit][0] != pos:
            hit = prev[hit][0]
        return prev[hit][1]

    for _ in range(T):
        st = env.state
        joint = []
        for i, pl in enumerate(st.players):
            if rng.random() < eps:
                joint.append(int(rng.integers(0, 6)))
            else:
                joint.append(plan(pl, st, {q.position for j, q in enumerate(st.players) if j != i}))
        out.append(joint)
        _, _, _, done, _ = env.n_step([np.int64(a) for a in joint])
        if done:
            env.n_reset()
    return np.array(out, dtype=np.int32).reshape(T, P)


# ----------------------------------------------------------------------------- fixtures
def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}  ({os.path.getsize(path)/1024:.1f} KiB)")


def main():
    oenv, omdp, venv = import_reference()
    N, S_, E, Wd, ST, I = 0, 1, 2, 3, 4, 5

    # 1) layout parameters --------------------------------------------------
    params = {}
    custom_path = {n: os.path.join(HERE, "layouts", n + ".layout") for n in CUSTOM}
    LAYOUT_PATHS.update(custom_path)
    for name in LAYOUTS + CUSTOM:
        d = oenv.get_base_layout_params(custom_path.get(name, name), 200)
        params[name] = {k: (list(map(int, v)) if isinstance(v, (list, tuple)) else int(v)) for k, v in d.items()}
    with open(os.path.join(HERE, "layout_params.json"), "w") as f:
        json.dump(params, f, indent=1, sort_keys=True)
    print("  wrote layout_params.json")

    # 2)+6) per-layout rollouts: uniform random (1000 steps, SURVEY 8(c) hash) and guided (600 steps)
    hashes = {}
    for name in LAYOUTS + CUSTOM:
        P = params[name]["num_players"]
        rng = np.random.default_rng(0)
        acts = np.stack([rng.integers(0, 6, size=P) for _ in range(1000)]).astype(np.int32)
        rec = rollout(oenv, omdp, name, acts)
        h = hashlib.sha256()
        h.update(rec["init_obs"].tobytes())
        for t in range(1000):
            h.update(rec["obs"][t].tobytes())
            h.update(np.int32(rec["reward"][t]).tobytes())
            h.update(np.uint8(rec["done"][t]).tobytes())
        hashes[name] = h.hexdigest()[:32]
        save(f"random_{name}.npz", **rec)
        g = guided_actions(oenv, omdp, name, 600, seed=1)
        rec = rollout(oenv, omdp, name, g)
        print(f"    guided {name}: total reward {int(rec['reward'].sum())}, "
              f"deliveries {(rec['reward_per_player'] >= 20).sum()}")
        save(f"guided_{name}.npz", **rec)
    with open(os.path.join(HERE, "random_rollout_sha256.json"), "w") as f:
        json.dump(hashes, f, indent=1, sort_keys=True)
    print("  rollout hashes:", hashes)

    # 3) scripted full recipe on `simple`: total reward 37
    p1 = [E, I, Wd, N, I] * 3 + [E]
    p0 = [ST] * 5 + [S_, I, N] + [ST] * 8 + [E, N] + [ST] * 16 + [I, S_, E, S_, I]
    T = max(len(p0), len(p1))
    p0 += [ST] * (T - len(p0)); p1 += [ST] * (T - len(p1))
    rec = rollout(oenv, omdp, "simple", np.array(list(zip(p0, p1)), np.int32))
    assert int(rec["reward"].sum()) == 37, rec["reward"].sum()
    save("scripted_recipe_simple.npz", **rec)

    # 4)+5) micro known-answer cases: explicit start states, 1-3 steps each.
    def st(layout, tiles=None, players=None, timestep=0):
        env = oenv.SimplifiedOvercooked(LAYOUT_PATHS.get(layout, layout))
        t, p, ts = dump_state(env.state)
        t = t.copy(); p = p.copy()
        for pos, tup in (tiles or {}).items():
            t[pos] = tup
        for i, tup in (players or {}).items():
            p[i] = tup
        return t, p, np.int32(timestep)

    ON, TO, DI, SO = 2, 1, 3, 4
    cases = {
        # name: (layout, init, actions)
        "collide_same_target": ("simple", None, [[N, Wd]]),
        "collide_blocked_EW": ("simple", None, [[E, Wd]]),
        "collide_E_stay": ("simple", None, [[E, ST]]),
        "collide_stay_W": ("simple", None, [[ST, Wd]]),
        "walls_J": ("simple", None, [[Wd, E], [ST, ST]]),
        "follow_K": ("simple", st("simple", players={0: (6, 0, 0, 0, 0, 0), 1: (7, 0, 0, 0, 0, 0)}), [[E, E]]),
        "swap_blocked": ("simple", st("simple", players={0: (6, 0, 0, 0, 0, 0), 1: (7, 0, 0, 0, 0, 0)}), [[E, Wd]]),
        "counter_onion_B": ("simple", st("simple", players={0: (6, 0, ON, 0, 0, -1)}), [[I, ST], [I, ST]]),
        "counter_dish": ("simple", st("simple", players={0: (6, 0, DI, 0, 0, -1)}), [[I, ST], [ST, ST], [I, ST]]),
        "counter_soup_H": ("simple", st("simple", players={0: (6, 0, SO, 3, 0, 20)}), [[I, ST], [S_, ST], [N, ST], [I, ST]]),
        "counter_occupied": ("simple", st("simple", tiles={1: (DI, 0, 0, -1)}, players={0: (6, 0, ON, 0, 0, -1)}), [[I, ST]]),
        "dish_not_useful_C": ("simple", None, [[S_, ST], [I, ST]]),
        "dish_useful": ("simple", st("simple", tiles={2: (SO, 3, 0, 4)}), [[S_, ST], [I, ST]]),
        "dish_on_counter_blocks_reward": ("simple", st("simple", tiles={2: (SO, 3, 0, 4), 4: (DI, 0, 0, -1)}), [[S_, ST], [I, ST]]),
        "dish_both_players_order": ("unident_s", st("unident_s", tiles={22: (SO, 1, 0, -1)},
                                                     players={0: (30, 1, 0, 0, 0, 0), 1: (32, 1, 0, 0, 0, 0)}), [[I, I]]),
        "dish_two_pots_both_useful": ("unident_s", st("unident_s", tiles={22: (SO, 1, 0, -1), 31: (SO, 3, 0, 7)},
                                                       players={0: (30, 1, 0, 0, 0, 0), 1: (32, 1, 0, 0, 0, 0)}), [[I, I]]),
        "pot_full_D": ("simple", st("simple", tiles={2: (SO, 3, 0, 5)}, players={1: (7, 0, ON, 0, 0, -1)}), [[ST, I]]),
        "pot_dish_unready_E": ("simple", st("simple", tiles={2: (SO, 3, 0, 5)}, players={1: (7, 0, DI, 0, 0, -1)}), [[ST, I]]),
        "horizon_F": ("simple", st("simple", tiles={2: (SO, 2, 0, -1)}, players={0: (6, 2, ON, 0, 0, -1)}, timestep=199), [[ST, I]]),
        "tomato_pot_G": ("simple", st("simple", players={1: (7, 0, TO, 0, 0, -1)}), [[ST, I]]),
        "soup_in_pot_I": ("simple", st("simple", tiles={2: (SO, 3, 0, 20)}), [[ST, ST], [ST, ST]]),
        "soup_pickup": ("simple", st("simple", tiles={2: (SO, 3, 0, 20)}, players={1: (7, 0, DI, 0, 0, -1)}), [[ST, I]]),
        "soup_almost_ready": ("simple", st("simple", tiles={2: (SO, 3, 0, 19)}, players={1: (7, 0, DI, 0, 0, -1)}), [[ST, I], [ST, I]]),
        "both_add_same_pot": ("unident_s", st("unident_s", tiles={22: (SO, 2, 0, -1)},
                                               players={0: (21, 2, ON, 0, 0, -1), 1: (23, 3, ON, 0, 0, -1)}), [[I, I], [ST, ST]]),
        "both_add_empty_pot": ("unident_s", st("unident_s",
                                                players={0: (21, 2, ON, 0, 0, -1), 1: (23, 3, TO, 0, 0, -1)}), [[I, I]]),
        "third_onion_autostart": ("simple", st("simple", tiles={2: (SO, 2, 0, -1)}, players={1: (7, 0, ON, 0, 0, -1)}), [[ST, I], [ST, ST]]),
        "serve": ("simple", st("simple", players={0: (13, 1, SO, 3, 0, 20)}), [[I, ST]]),
        "serve_mixed_recipe": ("simple", st("simple", players={0: (13, 1, SO, 1, 2, 20)}), [[I, ST]]),
        "serve_not_soup": ("simple", st("simple", players={0: (13, 1, DI, 0, 0, -1)}), [[I, ST]]),
        "onion_source_full_hands": ("simple", st("simple", players={0: (6, 3, DI, 0, 0, -1)}), [[I, ST]]),
        "onion_source": ("simple", st("simple", players={0: (6, 3, 0, 0, 0, 0)}), [[I, ST]]),
        "tomato_source": ("mdp_test", st("mdp_test", players={0: (11, 3, 0, 0, 0, 0)}), [[I, ST], [N, ST], [N, ST]]),
        "tomato_two_pots": ("mdp_test", st("mdp_test", tiles={2: (SO, 1, 1, -1), 17: (SO, 0, 2, -1)},
                                           players={0: (7, 0, TO, 0, 0, -1), 1: (12, 1, ON, 0, 0, -1)}), [[I, I], [ST, ST]]),
        "interact_air": ("simple", st("simple", players={0: (6, 2, ON, 0, 0, -1)}), [[I, ST]]),
    }
    for cname, (layout, init, acts) in cases.items():
        rec = rollout(oenv, omdp, layout, np.array(acts, np.int32), init=init)
        save(f"micro_{cname}.npz", layout=np.array(layout), **rec)

    # fuzz: arbitrary (also unreachable) states x one random joint action each.  Objects of any kind on any
    # tile, soups with any ingredient count / tick (cooking soups on counters, idle full soups, foreign
    # objects in pots), players anywhere walkable incl. the same tile, any held object, any timestep.
    for name in ("simple", "unident_s", "mdp_test", "random0", "six_pots", "tiny_mixed"):
        prm = params[name]
        S = prm["width"] * prm["height"]
        air = [i for i, t in enumerate(prm["terrain"]) if t == 0]
        Wd_ = prm["width"]
        facing = [(p_, o_) for p_ in air for o_, d_ in enumerate((-Wd_, Wd_, 1, -1))
                  if prm["terrain"][p_ + d_] in (1, 3, 4, 5, 6) or (prm["terrain"][p_ + d_] == 2 and o_ == 0)]
        rng = np.random.default_rng(sum(map(ord, name)) + 17)
        cases = dict(tiles=[], players=[], timestep=[], actions=[], n_tiles=[], n_players=[], n_timestep=[],
                     reward_per_player=[], done=[], obs=[], init_obs=[])

        def rand_obj():
            nm = int(rng.integers(1, 5))
            if nm == 4:
                return (4, int(rng.integers(0, 4)), int(rng.integers(0, 4)), int(rng.integers(-1, 26)))
            return (nm, 0, 0, -1)
        for _ in range(500):
            tiles = np.zeros((S, 4), np.int8)
            for pos in rng.choice(S, size=int(rng.integers(0, 7)), replace=False):
                tiles[pos] = rand_obj()
            for pos in [i for i, t in enumerate(prm["terrain"]) if t == 1]:
                if rng.random() < 0.7:      # pots mostly hold soups in every phase of cooking
                    tiles[pos] = (4, int(rng.integers(0, 4)), int(rng.integers(0, 2)), int(rng.choice([-1, -1, 0, 5, 19, 20, 21])))
            players = np.zeros((2, 6), np.int8)
            for i in range(2):
                held = rand_obj() if rng.random() < 0.6 else (0, 0, 0, 0)
                if rng.random() < 0.7:      # stand in front of something interactive, facing it
                    pos_or = facing[int(rng.integers(0, len(facing)))]
                else:
                    pos_or = (int(rng.choice(air)), int(rng.integers(0, 4)))
                players[i] = pos_or + held
            ts = int(rng.integers(0, 200)) if rng.random() < 0.9 else 199
            act = rng.integers(0, 6, size=(1, 2)).astype(np.int32)
            if rng.random() < 0.5:
                act[0, int(rng.integers(0, 2))] = 5          # bias towards INTERACT
            rec = rollout(oenv, omdp, name, act, init=(tiles, players, np.int32(ts)))
            cases["tiles"].append(tiles); cases["players"].append(players); cases["timestep"].append(ts)
            cases["actions"].append(act[0]); cases["init_obs"].append(rec["init_obs"])
            cases["n_tiles"].append(rec["tiles"][0]); cases["n_players"].append(rec["players"][0])
            cases["n_timestep"].append(rec["timestep"][0]); cases["reward_per_player"].append(rec["reward_per_player"][0])
            cases["done"].append(rec["done"][0]); cases["obs"].append(rec["obs"][0])
        save(f"fuzz_{name}.npz", **{k: np.array(v) for k, v in cases.items()})

    # a12) the batched surface: reference SyncVectorEnv over 3 reference envs, 420 steps (2 resets)
    import torch
    vec = venv.SyncVectorEnv([lambda: oenv.SimplifiedOvercooked("simple") for _ in range(3)], device="cpu")
    obs0 = vec.n_reset()
    rng = np.random.default_rng(7)
    acts = rng.integers(0, 6, size=(420, 2, 3)).astype(np.int32)
    vobs, vrew, vdone = [], [], []
    for t in range(420):
        obs, rews, dones, _ = vec.n_step(torch.from_numpy(acts[t]).long())
        vobs.append(torch.stack([o.obs for o in obs]).numpy().astype(np.int8))
        vrew.append(rews.numpy().astype(np.int32)); vdone.append(dones.numpy().astype(np.uint8))
    save("syncvec_simple.npz", actions=acts,
         init_obs=torch.stack([o.obs for o in obs0]).numpy().astype(np.int8),
         obs=np.stack(vobs), reward=np.stack(vrew), done=np.stack(vdone))


if __name__ == "__main__":
    main()
