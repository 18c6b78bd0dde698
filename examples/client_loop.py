"""A self-play client's main loop (what src_cpp/elfgames/go/train/distri_client.h does around the game threads), with the transport
left to the caller: games play on the GPU, their finished games and thread states collect in a ClientRecords (GuardedRecords), the
message is handed to `exchange(text) -> reply text`, and the server's reply (a MsgRequestSeq) is sent to the games.

    python examples/client_loop.py            # needs an MI355X; plays 9x9 games with a random-init net against a stand-in "server"

The stand-in server below answers like the reference's: a self-play request first, then an evaluation request written the way
EvalSubCtrl writes it (second AI, Dirichlet noise and the q_zero flags off), then self-play with the next model.
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

import elf_amd  # noqa: E402
from elf_amd.client import TsOptions  # noqa: E402
from elf_amd.selfplay import SpRequest  # noqa: E402


class StandInServer:
    """answers MsgRequestSeq texts; a real deployment sends the message over ZeroMQ to the reference's training server instead"""

    def __init__(self, ts):
        self.seq, self.ts, self.received = 0, ts, []

    def exchange(self, text):
        msg = json.loads(text)
        self.received.append((len(msg.get("records", [])), len(msg.get("states", []))))
        self.seq += 1
        ts = TsOptions.from_buffer_copy(self.ts)
        if self.seq == 2:      # an evaluation request (train/ctrl_eval.h:227-237,352-362)
            ts.root_epsilon = ts.root_alpha = 0.0
            ts.unexplored_q_zero = ts.root_unexplored_q_zero = 0
            q = SpRequest(2, 1, 0.05, 0.05, 0.0, -1, 0, 0, 2)
        else:
            q = SpRequest(1 if self.seq < 2 else 2, -1, 0.05, 0.05, 0.1, -1, 0, 0, 1)
        return elf_amd.request_seq_to_json(q, ts, self.seq)


def main():
    n, games = 9, 16
    dev = torch.device("cuda", 0)
    ts = TsOptions(0, 1, 32, 16, 0, 0, 1, 0, 0, 0.25, 0.03, 1, 1, 0, 0, 1.5, b"")
    sp = elf_amd.SelfPlay(board_size=n, num_games=games, mcts_rollout_per_thread=32, mcts_rollout_per_batch=16, move_cutoff=20,
                          keep_records=64, nodes_per_game=1024, model_ver=1)
    from elf_amd.net import make_net
    net = make_net(board_size=n, num_block=2, dim=32, device=dev, dtype=torch.float32, channels_last=False)
    server = StandInServer(ts)
    out = elf_amd.ClientRecords("example-client")
    versions = {"black": 1, "white": -1}
    import ctypes as C
    L = elf_amd.lib()
    bv, wv = C.c_int64(0), C.c_int64(0)
    for step in range(600):
        rb, rw = sp.begin_step2()
        if L.elfsp_take_game_starts(sp._h, C.byref(bv), C.byref(wv)):      # the "game_start" batch: load the models it names
            versions = {"black": bv.value, "white": wv.value}
        replies = [None, None]
        for a, (rows, s, ver) in enumerate(((rb, sp.s, versions["black"]), (rw, getattr(sp, "s_white", None), versions["white"]))):
            if rows:
                with torch.no_grad():
                    r = net(s[:rows])
                replies[a] = (r["pi"], r["V"], torch.full((rows,), ver, dtype=torch.int64, device=dev))
        sp.end_step2(replies)
        out.update_from(sp)
        if step % 100 == 99:
            request, seq, mcts_opt = elf_amd.parse_request_seq(server.exchange(out.dump_and_clear()))
            sp.send_request(request, mcts_opt)
            print("step", step, "sent", server.received[-1], "-> request", request.black_ver, request.white_ver, "seq", seq,
                  "root_epsilon", mcts_opt.root_epsilon)
    print("messages (records, states):", server.received)


if __name__ == "__main__":
    main()
