"""Random walled kitchens for differential tests (own generator; not reference data)."""
import numpy as np


def random_layout(rng, max_w=10, max_h=7):
    """-> flat params dict like get_base_layout_params: walled border, random interior counters, at least one
    of every station on the border, two distinct free start tiles, random recipe tables and shaped rewards."""
    while True:
        W, H = int(rng.integers(4, max_w + 1)), int(rng.integers(3, max_h + 1))
        if W * H > 255:
            continue
        terrain = np.zeros((H, W), dtype=int)
        terrain[0, :] = terrain[-1, :] = terrain[:, 0] = terrain[:, -1] = 2          # counters all around
        interior = [(y, x) for y in range(1, H - 1) for x in range(1, W - 1)]
        for (y, x) in interior:
            if rng.random() < 0.15:
                terrain[y, x] = int(rng.choice([2, 2, 1, 3, 4]))                     # islands: counters, pots, dispensers
        border = [(y, x) for y in range(H) for x in range(W) if (y in (0, H - 1)) != (x in (0, W - 1))]
        rng.shuffle(border)
        stations = [1, 3, 4, 5, 6] + [int(rng.integers(1, 7)) for _ in range(int(rng.integers(0, 6)))]
        for code, (y, x) in zip(stations, border):
            terrain[y, x] = code
        free = [(y, x) for (y, x) in interior if terrain[y, x] == 0]
        if len(free) < 2:
            continue
        picks = rng.choice(len(free), size=2, replace=False)
        starts = [free[int(i)] for i in picks]
        uniform = rng.random() < 0.5
        times = [int(rng.integers(1, 12))] * 16 if uniform else [int(v) for v in rng.integers(0, 15, size=16)]
        return {
            "terrain": [int(v) for v in terrain.reshape(-1)], "height": H, "width": W, "num_players": 2,
            "start_player_x": [int(s[1]) for s in starts], "start_player_y": [int(s[0]) for s in starts],
            "placement_in_pot_rew": int(rng.integers(0, 6)), "dish_pickup_rew": int(rng.integers(0, 6)),
            "soup_pickup_rew": int(rng.integers(0, 9)),
            "recipe_values": [int(v) for v in rng.integers(0, 40, size=16)], "recipe_times": times,
            "horizon": int(rng.integers(20, 120)),
        }


def busy_actions(rng, T, n):
    """Uniform actions biased towards INTERACT so that dispensers / pots / counters are really used."""
    a = rng.integers(0, 6, size=(T, 2, n))
    a[rng.random((T, 2, n)) < 0.25] = 5
    return a.astype(np.int32)


def random_layout_players(rng, players, max_w=10, max_h=7):
    """The same kitchens with 1, 3 or 4 players: `random_layout` (its random stream untouched, so the 2-player seeds keep their
    kitchens), then the first start dropped or further distinct free tiles drawn as start positions."""
    while True:
        p = random_layout(rng, max_w, max_h)
        W, H = p["width"], p["height"]
        starts = list(zip(p["start_player_y"], p["start_player_x"]))
        free = [(y, x) for y in range(1, H - 1) for x in range(1, W - 1) if p["terrain"][y * W + x] == 0 and (y, x) not in starts]
        if players > 2 and len(free) < players - 2:
            continue
        if players < 2:
            starts = starts[:players]
        else:
            extra = rng.choice(len(free), size=players - 2, replace=False) if players > 2 else []
            starts += [free[int(i)] for i in extra]
        p = dict(p, num_players=players, start_player_x=[int(s[1]) for s in starts], start_player_y=[int(s[0]) for s in starts])
        return p


def busy_actions_players(rng, T, players, n):
    a = rng.integers(0, 6, size=(T, players, n))
    a[rng.random((T, players, n)) < 0.25] = 5
    return a.astype(np.int32)
