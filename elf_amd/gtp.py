"""GTP front-end over the device-resident engine.

Mirrors the reference's console: `GoConsoleGTP` (scripts/elfgames/go/console_lib.py:207-372) driven by `df_console.py`, i.e. the
`human_actor` batch group of GoGameSelfPlay::act (src_cpp/elfgames/go/common/game_selfplay.cc:290-330): a human move is forwarded
on the game board, `genmove` lets the MCTS AI search and play, `clear_board` finishes the game (FR_CLEAR).  Same command set,
same replies ("= ..." / "? ..."), same coordinate letters (no 'I'): protocol_version, name, version, komi, boardsize,
clear_board, play, genmove, showboard, final_score, list_commands, quit/exit.

    eng = GtpEngine(actor, board_size=19, mcts_rollout_per_thread=1600)     # actor(batch) -> dict(pi=..., V=...)
    eng.loop()                                                               # stdin/stdout, or eng.command("genmove b")
"""
import inspect
import sys

from .engine import M_PASS

M_RESIGN = 1   # base/common.h:44
from .selfplay import SelfPlay


def move2xy(v):
    """console_lib.py:12-20"""
    if v.lower() == "pass":
        return -1, -1
    x = ord(v[0].lower()) - ord("a")
    if x >= 9:      # skip 'i'
        x -= 1
    y = int(v[1:]) - 1
    return x, y


def xy2move(x, y):
    """console_lib.py:23-29"""
    if x == -1 and y == -1:
        return "pass"
    if x >= 8:
        x += 1
    return chr(x + 65) + str(y + 1)


class GtpEngine:
    def __init__(self, actor, board_size=19, komi=7.5, device=0, **selfplay_options):
        opts = dict(mcts_rollout_per_thread=1600, mcts_rollout_per_batch=8, mcts_puct=1.5, mcts_virtual_loss=1,
                    mcts_persistent_tree=True, policy_distri_cutoff=0, resign_thres=0.0, seed=1)
        opts.update(selfplay_options)
        self.n = int(board_size)
        self.komi = float(komi)
        self.sp = SelfPlay(board_size=self.n, num_games=1, device=device, komi=self.komi, **opts)
        self.sp.reg_callback("actor_black", actor)
        self.boards = self.sp.board_engine()
        self.exit = False
        self.commands = {k[3:]: f for k, f in inspect.getmembers(self, predicate=inspect.ismethod) if k.startswith("on_")}

    def close(self):
        self.sp.close()

    # ---- board queries (GoGameSelfPlay.showBoard/getNextPlayer/getLastMove/getScore, inference/Pybind.cc:31-45)
    def _info(self):
        return self.boards.info_host(n=1)

    def next_player(self):
        return "B" if int(self._info()["next_player"][0]) == 1 else "W"

    def coord2move(self, c):
        if c == M_PASS:
            return "pass"
        S = self.n + 2
        return xy2move(c % S - 1, c // S - 1)

    def move2coord(self, v):
        x, y = move2xy(v)
        if (x, y) == (-1, -1):
            return M_PASS
        if not (0 <= x < self.n and 0 <= y < self.n):
            raise ValueError("off board")
        return (y + 1) * (self.n + 2) + (x + 1)

    def showboard(self):
        col, _ = self.boards.export_board(n=1)
        col = col.cpu().numpy()[0].reshape(self.n, self.n)     # [x][y]
        letters = [xy2move(x, 0)[0] for x in range(self.n)]
        rows = ["   " + " ".join(letters)]
        for y in range(self.n - 1, -1, -1):
            rows.append("%2d " % (y + 1) + " ".join(".XO"[int(col[x, y])] for x in range(self.n)) + " %d" % (y + 1))
        rows.append("   " + " ".join(letters))
        info = self._info()
        rows.append("Next: %s  ply %d  captures B %d W %d" % (self.next_player(), int(info["ply"][0]), int(info["b_cap"][0]), int(info["w_cap"][0])))
        return "\n".join(rows)

    def check_player(self, player):
        """console_lib.py:310-322"""
        nxt = self.next_player()
        if player.lower() != nxt.lower():
            return False, "Specified next player %s is not the same as the next player %s on the board" % (player, nxt)
        return True, None

    # ---- GTP commands (console_lib.py:208-282)
    def on_protocol_version(self, items):
        return True, "2"

    def on_name(self, items):
        return True, "DF2"

    def on_version(self, items):
        return True, "1.0"

    def on_komi(self, items):
        if float(items[1]) != self.komi:
            return False, "We only support %g komi for now" % self.komi
        return True, None

    def on_boardsize(self, items):
        if items[1] != str(self.n):
            return False, "We only support %dx%d board for now" % (self.n, self.n)
        return True, None

    def on_clear_board(self, items):
        # M_CLEAR of the human actor (game_selfplay.cc:306-311): a game that has not started yet is left alone
        if int(self._info()["ply"][0]) > 1:
            self.sp.restart([0])
        return True, None

    def on_play(self, items):
        ret, msg = self.check_player(items[1][0])
        if not ret:
            return False, msg
        try:
            c = self.move2coord(items[2])
            self.sp.play([c])
        except Exception:
            return False, "illegal move"
        return True, None

    def on_genmove(self, items):
        ret, msg = self.check_player(items[1][0])
        if not ret:
            return False, msg
        moves = self.sp.stats()["moves"]
        while self.sp.stats()["moves"] == moves:
            self.sp.run()
        # what the search did -- not inferred from the game counter: the engine may have resigned (no move, board restarted), or
        # its move may have ended the game (two passes / move limit), in which case the board shows the next game already
        c = int(self.sp.last_moves()[0])
        if c == M_RESIGN:
            return True, "resign"
        return True, self.coord2move(c)

    def on_showboard(self, items):
        return True, "\n" + self.showboard()

    def on_final_score(self, items):
        if int(self._info()["ply"][0]) > 1:
            score = float(self.boards.evaluate(komi=self.komi, n=1).cpu()[0])     # GoGameSelfPlay::getScore
        else:
            score = float(self.sp.last_score()[0])                                # getLastScore
        return True, ("B+%.1f" % score) if score > 0 else ("W+%.1f" % -score)

    def on_list_commands(self, items):
        return True, "\n".join(self.commands.keys())

    def on_quit(self, items):
        self.exit = True
        return True, None

    def on_exit(self, items):
        return self.on_quit(items)

    # ---- protocol
    def command(self, line):
        """One GTP command line -> the reply text ("= ..." or "? ..."), console_lib.py:324-372"""
        items = line.split()
        if not items:
            return "? Invalid input\n\n"
        try:
            ret, msg = self.commands[items[0]](items)
        except KeyError:
            return "? unknown command\n\n"
        except Exception as e:
            return "? Invalid command (%s)\n\n" % e
        return "%s %s\n\n" % ("=" if ret else "?", msg if msg is not None else "")

    def loop(self, fin=sys.stdin, fout=sys.stdout):
        for line in fin:
            fout.write(self.command(line.strip()))
            fout.flush()
            if self.exit:
                break
