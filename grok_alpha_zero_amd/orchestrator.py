"""Generation loop around the engine — the self-play half of the reference's `Run()` (<Game>/main.py:232-352).

    Run(game_class, configs, train_fn, weights_fn)

keeps the reference's on-disk contract (`Grok_Zero_Train/<generation>/Self_Play_Data.h5`, resume from the highest generation
folder and from `game_stats[2]` inside it, `Print_Stats`, `compute_speed`) and replaces `run_self_play`'s worker processes +
inference server by one engine per GPU.  Training, ONNX export and TensorRT caching are NOT part of this path (SURVEY §2.1:
out of scope): the caller supplies

    train_fn(generation, folder_path, save_folder_path)   # Train_NN of <Game>/main.py:99-136; must leave the next
                                                          # generation's weights where weights_fn finds them
    weights_fn(folder_path) -> dict | None                # tensors for SelfPlayEngine.load_weights (net.export_engine_weights()
                                                          # or keras_weights.load_keras_weights); None = synthetic evaluator

Generation 0 plays with the synthetic evaluator, like the reference's session=None dummy (Self_Play.py:40).
"""
import os
import time
from glob import glob
from pathlib import Path

import numpy as np

from .self_play import ReplayStore, run_self_play


def make_generation_folder(root, generation):
    os.makedirs(os.path.join(root, str(generation)), exist_ok=True)                     # <Game>/main.py:77-80


def make_dataset_file(folder_path):
    """`Self_Play_Data.h5` with a zeroed `game_stats` u32[6] (<Game>/main.py:82-86)."""
    ReplayStore(folder_path).create()


def current_generation(root):
    """Highest numbered folder under `root`, 0 when there is none (<Game>/main.py:251-254)."""
    gens = [int(Path(p).name) for p in glob(os.path.join(root, "*")) if Path(p).name.isdigit()]
    return max(gens) if gens else 0


def print_stats(folder_path, out=print):
    """<Game>/main.py:88-96."""
    max_actions, total_actions, n_games, p1, draws, p2 = (int(x) for x in ReplayStore(folder_path).game_stats())
    n = max(n_games, 1)
    out("---------Game Statistics---------")
    out(f"Longest game is: {max_actions} actions long!")
    out(f"Average moves: {round(total_actions / n, 4)}")
    out(f"Player -1 winrate: {round(p1 / n, 4)}")
    out(f"Draw rate: {round(draws / n, 4)}")
    out(f"Player 1 winrate: {round(p2 / n, 4)}\n")
    return dict(max_actions=max_actions, total_actions=total_actions, games=n_games, wins_m1=p1, draws=draws, wins_p1=p2)


def compute_speed(game_class, configs, weights, batch=None, iterations=200, device=0, lib_path=None, out=print):
    """Evaluator probe (Compute_Speed.py:9-63): forward passes per second at `batch` positions (default `num_workers`, the
    reference's batch; the engine's own batch is the number of concurrent games).  Returns it/s."""
    from .engine import EVAL_RESNET, SelfPlayEngine
    build_config, train_config = configs[0], configs[1]
    name = getattr(game_class, "ENGINE_NAME", game_class.__name__)
    batch = int(batch or train_config.get("num_workers", 1))
    eng = SelfPlayEngine(name, batch, 1, train_config["max_actions"], 0, 0, 0.0, 0.0, 0, evaluator=EVAL_RESNET,
                         net_blocks=build_config["num_resnet_layers"], net_filters=build_config.get("num_filters", 128),
                         ring_capacity=0, device=device, lib_path=lib_path)
    try:
        eng.load_weights(weights)
        shape = game_class().get_input_state().shape
        x = np.random.default_rng(0).integers(-1, 2, size=(batch, *shape)).astype(np.int8)
        _, _, ms = eng.evaluate(x, repeats=iterations)
    finally:
        eng.close()
    out(f"Took {ms / 1e3} seconds per iteration at {1e3 / ms:0.2f} it/s!")
    return 1e3 / ms


def Run(game_class, configs, train_fn=None, weights_fn=None, root="Grok_Zero_Train", n_games=1024, seed=None, device=0,
        lib_path=None, out=print, self_play_kwargs=None):
    """The loop of <Game>/main.py:312-352: for every generation self-play (resumable), statistics, `train_fn`, next dataset file.
    Returns the list of per-generation statistics."""
    train_config = configs[1]
    total = int(train_config["total_generations"])
    gen = current_generation(root)
    if gen == 0 and not ReplayStore(os.path.join(root, "0")).exists():                  # Initialize (<Game>/main.py:255-269)
        make_generation_folder(root, 0)
        make_dataset_file(os.path.join(root, "0"))
    out("\n*************Starting*************\n")
    out(f"Generation: {gen} / {total - 1}")
    if not ReplayStore(os.path.join(root, str(gen))).exists():                          # training of gen-1 finished, file missing
        make_dataset_file(os.path.join(root, str(gen)))
    stats = []
    for generation in range(gen, total):
        folder = os.path.join(root, str(generation))
        weights = weights_fn(folder) if (weights_fn and generation > 0) else None
        t0 = time.time()
        played = run_self_play(game_class, configs, folder, n_games=n_games, seed=None if seed is None else seed + generation,
                               weights=weights, device=device, lib_path=lib_path, **(self_play_kwargs or {}))
        st = print_stats(folder, out)
        st.update(generation=generation, played_now=played, seconds=time.time() - t0)
        stats.append(st)
        if train_fn is not None:
            make_generation_folder(root, generation + 1)
            train_fn(generation, folder, os.path.join(root, str(generation + 1)))
        if generation < total - 1:
            make_generation_folder(root, generation + 1)
            make_dataset_file(os.path.join(root, str(generation + 1)))
            out(f"Generation: {generation + 1} / {total - 1}")
    out("-----------Training Done!-----------")
    return stats
