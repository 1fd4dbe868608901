"""End-to-end run_self_play on the GPU: engine + record conversion + augmentation + HDF5 replay writer, games per second."""
import os, sys, time, tempfile, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grok_alpha_zero_amd.games import GAMES
from grok_alpha_zero_amd.net import Connect4Net
from grok_alpha_zero_amd.self_play import ReplayStore, run_self_play
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
root = tempfile.mkdtemp(); folder = os.path.join(root, "1"); ReplayStore(folder).create()
train = dict(games_per_generation=n, MCTS_iteration_limit=134, max_actions=42, num_explore_actions_first=8, num_explore_actions_second=7,
             c_puct_init=2.5, dirichlet_alpha=0.5, use_gumbel=False)
build = dict(num_resnet_layers=6, num_filters=128)
w = Connect4Net(6).eval().export_engine_weights()
pr = cProfile.Profile(); t0 = time.time(); pr.enable()
played = run_self_play(GAMES["Connect4"], (build, train), folder, n_games=4096, seed=1, weights=w)
pr.disable(); dt = time.time() - t0
gs = ReplayStore(folder).game_stats()
print(f"{played} games, {int(gs[1])} positions in {dt:.1f} s = {played / dt:.0f} games/s, {int(gs[1]) / dt:.0f} positions/s; file {os.path.getsize(os.path.join(folder, 'Self_Play_Data.h5')) / 1e6:.0f} MB")
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
