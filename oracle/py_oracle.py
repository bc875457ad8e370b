"""Pure-Python restatement of the reference's single-env CPU path -- TEST INFRASTRUCTURE ONLY.

Second, independent oracle (the first is overcooked_oracle.c): plain lists and ints, one kitchen at a
time, written from the transition / encoding spec in SURVEY.md 8(a) and checked line by line against
/root/reference/envs/overcooked2_reimplement.py (DummyMDP, "ref:" numbers) and
envs/overcooked2_env.py:319-343 (SimplifiedOvercooked).  It is the counterpart of BASELINE config 1
("cramped_room, 1 env, reference Python step on CPU"): bench.py times it on the GPU box's host, where the
reference itself cannot run.  Parity: pinned by tests/test_oracle_golden.py on every fixture.

State uses the fixture serialisation: tiles [S][4], players [P][6], timestep.
"""
NONE, TOMATO, ONION, DISH, SOUP = range(5)
AIR, POT, COUNTER, ONION_SRC, DISH_SRC, SERVING, TOMATO_SRC = range(7)
NORTH, SOUTH, EAST, WEST, STAY, INTERACT = range(6)
EMPTY = (0, 0, 0, 0)


class PyKitchen:
    def __init__(self, params):
        self.p = params
        self.W, self.H = params["width"], params["height"]
        self.S, self.P = self.W * self.H, params["num_players"]
        self.terrain = list(params["terrain"])
        self.delta = {NORTH: -self.W, SOUTH: self.W, EAST: 1, WEST: -1, STAY: 0}           # ref:22-32
        self.C = 5 * self.P + 10
        self.reset()

    # ref:387-391
    def reset(self):
        self.tiles = [EMPTY] * self.S
        self.players = [[y * self.W + x, NORTH, EMPTY] for x, y in zip(self.p["start_player_x"], self.p["start_player_y"])]
        self.timestep = 0

    def load(self, tiles, players, timestep):
        self.tiles = [tuple(int(v) for v in t) for t in tiles]
        self.players = [[int(p[0]) & 0xFF, int(p[1]), tuple(int(v) for v in p[2:6])] for p in players]
        self.timestep = int(timestep)

    def dump(self):
        return ([list(t) for t in self.tiles], [[p[0], p[1], *p[2]] for p in self.players], self.timestep)

    def _time(self, o):                               # ref:56-57,156-157
        return self.p["recipe_times"][4 * o[1] + o[2]]

    def _ready(self, o):                              # ref:162-163
        return o[3] >= 0 and o[3] >= self._time(o)

    def _cooking(self, o):                            # ref:159-160
        return 0 <= o[3] < self._time(o)

    def transition(self, actions):
        """ref:381-385; returns the per-player reward list."""
        T, tiles, players = self.terrain, self.tiles, self.players
        # ref:272-281, taken once (ref:302)
        pots = sum(1 for i in range(self.S) if T[i] == POT and tiles[i][0] != NONE and (tiles[i][3] >= 0 or tiles[i][1] + tiles[i][2] < 3))
        rew = [0] * self.P
        for i, (pl, a) in enumerate(zip(players, actions)):                                  # ref:305-353
            if a != INTERACT:
                continue
            ipos = pl[0] + self.delta[pl[1]]
            terr, held, t = T[ipos], pl[2], tiles[ipos]
            if terr == COUNTER:
                if held[0] != NONE and t[0] == NONE:
                    tiles[ipos], pl[2] = held, EMPTY
                elif held[0] == NONE and t[0] != NONE:
                    pl[2], tiles[ipos] = t, EMPTY
            elif terr == ONION_SRC and held[0] == NONE:
                pl[2] = (ONION, 0, 0, -1)
            elif terr == TOMATO_SRC and held[0] == NONE:
                pl[2] = (TOMATO, 0, 0, -1)
            elif terr == DISH_SRC and held[0] == NONE:
                if self.P == 2:                                                              # ref:261-270
                    dishes = sum(1 for q in players if q[2][0] == DISH)
                    on_counter = any(T[j] == COUNTER and tiles[j][0] == DISH for j in range(self.S))
                    if not on_counter and dishes < pots:
                        rew[i] += self.p["dish_pickup_rew"]
                pl[2] = (DISH, 0, 0, -1)
            elif terr == POT and held[0] != NONE:
                if held[0] == DISH and t[0] != NONE and self._ready(t):
                    pl[2], tiles[ipos] = t, EMPTY
                    rew[i] += self.p["soup_pickup_rew"]
                elif held[0] in (ONION, TOMATO):
                    soup = list(t if t[0] != NONE else (SOUP, 0, 0, -1))
                    if not (soup[3] >= 0 or soup[1] + soup[2] == 3):
                        soup[1 if held[0] == ONION else 2] += 1
                        pl[2] = EMPTY
                        rew[i] += self.p["placement_in_pot_rew"]
                    s = tuple(soup)
                    if s[0] == SOUP and not self._cooking(s) and not self._ready(s) and s[1] + s[2] > 0 and s[1] + s[2] == 3:
                        s = (s[0], s[1], s[2], 0)                                            # ref:348
                    tiles[ipos] = s
            elif terr == SERVING and held[0] == SOUP:
                rew[i] += self.p["recipe_values"][4 * held[1] + held[2]]
                pl[2] = EMPTY
        # ref:368-371,393-399
        prop = []
        for pl, a in zip(players, actions):
            if a == INTERACT:
                prop.append((pl[0], pl[1]))
            else:
                new = pl[0] + self.delta[a]
                prop.append((pl[0] if T[new] != AIR else new, pl[1] if a == STAY else a))
        # ref:356-366
        blocked = any(prop[i][0] == prop[j][0] or (prop[i][0] == players[j][0] and players[i][0] == prop[j][0])
                      for i in range(self.P) for j in range(i + 1, self.P))
        for pl, (np_, no) in zip(players, prop):
            if not blocked:
                pl[0] = np_
            pl[1] = no
        # ref:373-379
        self.timestep += 1
        for i in range(self.S):
            o = tiles[i]
            if o[0] == SOUP and self._cooking(o):
                tiles[i] = (o[0], o[1], o[2], o[3] + 1)
        return rew

    def encode(self):
        """ref:165-259 + envs/overcooked2_env.py:322-325 -> nested list [P][W][H][C]."""
        T, C, P, shift = self.terrain, self.C, self.P, 5 * self.P
        base = [[0] * C for _ in range(self.S)]
        for pos in range(self.S):
            if T[pos] > AIR:
                base[pos][T[pos] - 1 + shift] = 1
            o = self.tiles[pos]
            if o[0] == SOUP:
                if T[pos] == POT:
                    base[pos][shift + 5] = o[1]
                    base[pos][shift + 6] = max(o[3], 0)
                else:
                    base[pos][shift + 7] = 1
            elif o[0] == DISH:
                base[pos][shift + 8] = 1
            elif o[0] == ONION:
                base[pos][shift + 9] = 1
        out = []
        for v in range(P):
            grid = [row[:] for row in base]
            other = 1
            for i, pl in enumerate(self.players):
                row = grid[pl[0]]
                if i == v:
                    row[0] = 1
                    row[P + pl[1]] = 1
                else:
                    row[other] = 1
                    row[P + 4 * other + pl[1]] = 1
                    other += 1
                if pl[2][0] == SOUP:
                    row[shift + 7] = 1
                elif pl[2][0] == DISH:
                    row[shift + 8] = 1
                elif pl[2][0] == ONION:
                    row[shift + 9] = 1
            out.append([[grid[y * self.W + x] for y in range(self.H)] for x in range(self.W)])
        return out

    def n_step(self, actions):
        """SimplifiedOvercooked.n_step + the vector wrappers' reset rule -> (obs, shared reward, done)."""
        reward = sum(self.transition(actions))
        done = self.timestep >= self.p["horizon"]
        if done:
            self.reset()
        return self.encode(), reward, done
