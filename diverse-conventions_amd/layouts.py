"""Kitchen layouts: own data (`layouts/kitchens.json`) + the flat-parameter builder.

`get_base_layout_params(layout_name, horizon, max_num_players=None)` has the reference's name,
arguments and result (reference: envs/overcooked2_env.py:171-291): a dict with exactly the keyword
arguments of the native simulator constructor (reference bindings.cpp:28-84) --
terrain, height, width, num_players, start_player_x, start_player_y, placement_in_pot_rew,
dish_pickup_rew, soup_pickup_rew, recipe_values[16], recipe_times[16], horizon.
`layout_name` is a bare name from kitchens.json (old reference names or the classic Overcooked
aliases below) or a path to a reference-format `*.layout` file (a Python dict literal).
"""
import ast
import json
import os

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "layouts")

# the five classic layouts are stored under the reference's old names
ALIASES = {
    "cramped_room": "simple",
    "asymmetric_advantages": "unident_s",
    "coordination_ring": "random1",
    "forced_coordination": "random0",
    "counter_circuit": "random3",
}
CLASSIC_LAYOUTS = ["simple", "unident_s", "random1", "random0", "random3"]

MAX_NUM_INGREDIENTS = 3
N_RECIPES = (MAX_NUM_INGREDIENTS + 1) ** 2

# terrain codes (reference envs/overcooked2_reimplement.py:12-19, envs/overcooked2_env.py:145-152)
TERRAIN_CODE = {" ": 0, "P": 1, "X": 2, "O": 3, "D": 4, "S": 5, "T": 6}
# characters that mark player start tiles, in player order (reference envs/overcooked2_env.py:153-156)
PLAYER_MARKS = "1234567890!@#$%^&*()abcdefghijklmnopqrst"

DEFAULT_SHAPING = {"PLACEMENT_IN_POT_REW": 3, "DISH_PICKUP_REWARD": 3, "SOUP_PICKUP_REWARD": 5}

_kitchens = None


def _load_kitchens():
    global _kitchens
    if _kitchens is None:
        with open(os.path.join(_DIR, "kitchens.json")) as f:
            _kitchens = json.load(f)
    return _kitchens


def layout_names():
    return sorted(_load_kitchens())


def read_layout_dict(layout_name):
    """Raw layout description: {'grid': [row, ...] or a newline-separated string, 'cook_time', ...}."""
    if layout_name.endswith(".layout"):
        with open(layout_name) as f:
            return dict(ast.literal_eval(f.read()))       # dict literal; never eval()
    name = ALIASES.get(layout_name, layout_name)
    table = _load_kitchens()
    if name not in table:
        raise FileNotFoundError(f"unknown layout {layout_name!r}; known: {', '.join(layout_names())}")
    return dict(table[name])


def _recipe_index(order):
    ing = order["ingredients"]
    return (MAX_NUM_INGREDIENTS + 1) * sum(i == "onion" for i in ing) + sum(i == "tomato" for i in ing)


def _recipe_table(d, default, per_onion, per_tomato, listed, orders, uniform):
    """16-entry table indexed 4*onions+tomatoes, with the reference's override order
    (envs/overcooked2_env.py:236-273): default 20 < per-ingredient linear < per-order list < uniform."""
    table = [default] * N_RECIPES
    if per_onion in d and per_tomato in d:
        table = [o * d[per_onion] + t * d[per_tomato]
                 for o in range(MAX_NUM_INGREDIENTS + 1) for t in range(MAX_NUM_INGREDIENTS + 1)]
    if listed in d:
        for order, v in zip(orders, d[listed]):
            table[_recipe_index(order)] = v
    if uniform in d:
        table = [d[uniform]] * N_RECIPES
    return table


def get_base_layout_params(layout_name, horizon, max_num_players=None):
    d = read_layout_dict(layout_name)
    grid = d["grid"]
    rows = [r.strip() for r in (grid.split("\n") if isinstance(grid, str) else grid)]
    width = len(rows[0])
    if any(len(r) != width for r in rows):
        raise ValueError(f"layout {layout_name!r}: ragged grid rows")

    starts = {}
    terrain = []
    for y, row in enumerate(rows):
        for x, ch in enumerate(row):
            if ch in PLAYER_MARKS:
                idx = PLAYER_MARKS.index(ch)
                if max_num_players is None or idx < max_num_players:
                    starts[idx] = (x, y)
                ch = " "
            if ch not in TERRAIN_CODE:
                raise ValueError(f"layout {layout_name!r}: unknown tile {ch!r} at ({x},{y})")
            terrain.append(TERRAIN_CODE[ch])
    # players are the marks present, in mark order, truncated at the first gap like the reference's
    # fixed-size table (envs/overcooked2_env.py:190-199)
    n_players = len(starts)
    ordered = [starts[i] for i in sorted(starts)][:n_players]

    shaping = d.get("rew_shaping_params") or DEFAULT_SHAPING
    orders = d.get("start_all_orders") or []
    return {
        "terrain": terrain,
        "height": len(rows),
        "width": width,
        "num_players": n_players,
        "start_player_x": [p[0] for p in ordered],
        "start_player_y": [p[1] for p in ordered],
        "placement_in_pot_rew": shaping["PLACEMENT_IN_POT_REW"],
        "dish_pickup_rew": shaping["DISH_PICKUP_REWARD"],
        "soup_pickup_rew": shaping["SOUP_PICKUP_REWARD"],
        "recipe_values": _recipe_table(d, 20, "onion_value", "tomato_value", "recipe_values", orders, "delivery_reward"),
        "recipe_times": _recipe_table(d, 20, "onion_time", "tomato_time", "recipe_times", orders, "cook_time"),
        "horizon": horizon,
    }
