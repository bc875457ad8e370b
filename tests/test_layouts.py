"""Layout data + get_base_layout_params against the parameters the reference's parser produces
(tests/golden/layout_params.json, generated from envs/overcooked2_env.py:171-291)."""
import pytest

from diverse_conventions_amd import ALIASES, CLASSIC_LAYOUTS, get_base_layout_params, layout_names


def test_matches_reference_parser(layout_params):
    import os
    from conftest import GOLDEN
    for name, want in layout_params.items():
        custom = os.path.join(GOLDEN, "layouts", name + ".layout")     # this repo's own reference-format files
        arg = custom if os.path.exists(custom) else name
        assert get_base_layout_params(arg, 200) == want, name
        assert get_base_layout_params(arg, 77)["horizon"] == 77
    assert layout_params["six_pots"]["recipe_times"][4 * 2 + 1] == 11 and layout_params["six_pots"]["dish_pickup_rew"] == 7


def test_all_21_reference_layouts_present():
    assert len(layout_names()) == 21
    for n in CLASSIC_LAYOUTS:
        assert n in layout_names()


def test_classic_aliases(layout_params):
    for alias, old in ALIASES.items():
        assert get_base_layout_params(alias, 200) == layout_params[old]
    p = get_base_layout_params("cramped_room", 200)
    assert (p["width"], p["height"]) == (5, 4) and p["start_player_x"] == [1, 3] and p["start_player_y"] == [2, 1]


def test_reference_format_layout_file(tmp_path):
    f = tmp_path / "tiny.layout"
    f.write_text('''{
        "grid":  """XPXX
                    O 1X
                    X2 D
                    XSXX""",
        "start_order_list": None,
        "cook_time": 7,
        "num_items_for_soup": 3,
        "delivery_reward": 11,
        "rew_shaping_params": {"PLACEMENT_IN_POT_REW": 1, "DISH_PICKUP_REWARD": 2, "SOUP_PICKUP_REWARD": 4}
    }''')
    p = get_base_layout_params(str(f), 50)
    assert p["terrain"] == [2, 1, 2, 2, 3, 0, 0, 2, 2, 0, 0, 4, 2, 5, 2, 2]
    assert (p["start_player_x"], p["start_player_y"]) == ([2, 1], [1, 2])
    assert p["recipe_times"] == [7] * 16 and p["recipe_values"] == [11] * 16
    assert (p["placement_in_pot_rew"], p["dish_pickup_rew"], p["soup_pickup_rew"]) == (1, 2, 4)


def test_player_limit_and_errors(tmp_path):
    p = get_base_layout_params("multiplayer_schelling", 200)
    assert p["num_players"] == 4
    assert get_base_layout_params("multiplayer_schelling", 200, max_num_players=2)["num_players"] == 2
    with pytest.raises(FileNotFoundError):
        get_base_layout_params("nope", 200)
    bad = tmp_path / "bad.layout"
    bad.write_text('{"grid": """XXX\nX?X\nXXX"""}')
    with pytest.raises(ValueError):
        get_base_layout_params(str(bad), 200)


def test_per_ingredient_tables(tmp_path):
    f = tmp_path / "t.layout"
    f.write_text('{"grid": """XPX\nO1X\nX2S\nXDX""", "onion_time": 3, "tomato_time": 5, "onion_value": 2, "tomato_value": 7}')
    p = get_base_layout_params(str(f), 10)
    assert p["recipe_times"][4 * 2 + 1] == 2 * 3 + 5 and p["recipe_values"][4 * 1 + 3] == 2 + 21
