"""ctypes front-end of the CPU oracle (oracle/overcooked_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product
package.  State is held in the fixture serialisation (see the C file header)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# ORACLE_SANITIZE=1 runs the ASan/UBSan build of the oracle (CPU only; GPU sanitizers are not available on the pool):
#   ORACLE_SANITIZE=1 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python -m pytest tests/test_oracle_golden.py
SANITIZE = os.environ.get("ORACLE_SANITIZE", "0") not in ("", "0")
LIB_PATH = os.path.join(HERE, "liboracle_asan.so" if SANITIZE else "liboracle.so")
MAX_TILES, MAX_PLAYERS = 256, 8


class OcrLayout(ctypes.Structure):
    _fields_ = [
        ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("num_players", ctypes.c_int32),
        ("placement_in_pot_rew", ctypes.c_int32), ("dish_pickup_rew", ctypes.c_int32),
        ("soup_pickup_rew", ctypes.c_int32), ("horizon", ctypes.c_int32),
        ("recipe_values", ctypes.c_int32 * 16), ("recipe_times", ctypes.c_int32 * 16),
        ("start_x", ctypes.c_int32 * MAX_PLAYERS), ("start_y", ctypes.c_int32 * MAX_PLAYERS),
        ("terrain", ctypes.c_int32 * MAX_TILES),
    ]


def build(force=False):
    src = os.path.join(HERE, "overcooked_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B", os.path.basename(LIB_PATH)], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB_PATH)
        assert L.ocr_sizeof_layout() == ctypes.sizeof(OcrLayout)
        _lib = L
    return _lib


def make_layout(params):
    L = OcrLayout()
    for k in ("width", "height", "num_players", "placement_in_pot_rew", "dish_pickup_rew", "soup_pickup_rew", "horizon"):
        setattr(L, k, int(params[k]))
    for i in range(16):
        L.recipe_values[i] = int(params["recipe_values"][i])
        L.recipe_times[i] = int(params["recipe_times"][i])
    for i in range(params["num_players"]):
        L.start_x[i] = int(params["start_player_x"][i])
        L.start_y[i] = int(params["start_player_y"][i])
    for i, t in enumerate(params["terrain"]):
        L.terrain[i] = int(t)
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class OracleVecEnv:
    """N kitchens stepped on the CPU with the reference's semantics (auto-reset vector rule)."""

    def __init__(self, params, num_kitchens, nthreads=1):
        self.params = params
        self.L = make_layout(params)
        self.N = int(num_kitchens)
        self.P = params["num_players"]
        self.W, self.H = params["width"], params["height"]
        self.S = self.W * self.H
        self.C = 5 * self.P + 10
        self.nthreads = nthreads
        self.tiles = np.zeros((self.N, self.S, 4), np.int8)
        self.players = np.zeros((self.N, self.P, 6), np.int8)
        self.timestep = np.zeros((self.N,), np.int32)
        self.obs = np.zeros((self.P, self.N, self.W, self.H, self.C), np.int8)
        self.reward = np.zeros((self.P, self.N), np.int32)
        self.done = np.zeros((self.N,), np.int32)
        self.reset_all()

    def reset_all(self):
        lib().ocr_vec_start(ctypes.byref(self.L), ctypes.c_int64(self.N), _p(self.tiles), _p(self.players), _p(self.timestep))
        return self.encode()

    def encode(self):
        lib().ocr_vec_encode(ctypes.byref(self.L), ctypes.c_int64(self.N), _p(self.tiles), _p(self.players), _p(self.obs),
                             ctypes.c_int(self.nthreads))
        return self.obs

    def step(self, actions, want_obs=True):
        a = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.P, self.N)
        lib().ocr_vec_step(ctypes.byref(self.L), ctypes.c_int64(self.N), _p(self.tiles), _p(self.players), _p(self.timestep),
                           _p(a), _p(self.obs) if want_obs else None, _p(self.reward), _p(self.done),
                           ctypes.c_int(self.nthreads))
        return self.obs, self.reward, self.done

    def set_state(self, tiles, players, timestep, first=0):
        n = len(tiles)
        self.tiles[first:first + n] = tiles
        self.players[first:first + n] = players
        self.timestep[first:first + n] = timestep

    def get_state(self, first=0, count=None):
        count = self.N - first if count is None else count
        sl = slice(first, first + count)
        return self.tiles[sl].copy(), self.players[sl].copy(), self.timestep[sl].copy()


def transition(params, tiles, players, timestep, actions):
    """Single kitchen MDP transition (no wrapper): -> tiles, players, timestep, per-player rewards."""
    L = make_layout(params)
    t = np.ascontiguousarray(tiles, np.int8).copy()
    p = np.ascontiguousarray(players, np.int8).copy()
    ts = np.array([timestep], np.int32)
    a = np.ascontiguousarray(actions, np.int32)
    r = np.zeros((params["num_players"],), np.int32)
    lib().ocr_transition(ctypes.byref(L), _p(t), _p(p), _p(ts), _p(a), _p(r))
    return t, p, int(ts[0]), r


def encode_tile_major(params, tiles, players):
    """[P, S, C] exactly like the reference's lossless_state_encoding tuple."""
    L = make_layout(params)
    P, S = params["num_players"], params["width"] * params["height"]
    out = np.zeros((P, S, 5 * P + 10), np.int8)
    lib().ocr_encode(ctypes.byref(L), _p(np.ascontiguousarray(tiles, np.int8)), _p(np.ascontiguousarray(players, np.int8)), _p(out))
    return out
