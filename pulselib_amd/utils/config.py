"""Configuration files, read as the reference reads them (utils/config.py:6-15: `yaml.safe_load` of a file under
`config/`, None when it does not exist) and turned into the constructor arguments its entry point builds from them
(scripts/Poker/trainGPU.py:148-214).  The reference's own `config/pokerGPU.yaml` is consumable as it is: same flat keys,
same meaning, values passed through untouched (PyYAML reads `2e-4` as the STRING "2e-4"; the learner's constructor
floats it, Player.py:224-225 -- ours does the same).  Keys the reference's file carries but its entry point never reads
(CAPACITY, W1_DECAY, W2_DECAY, PLOT_FILENAME, SCORES_FILENAME) are accepted and ignored the same way.
PLOTTING is accepted and has NO effect: the reference's entry point builds a MatplotlibPlotter from it and writes
`rewards_learning_curve` / `total_chips_curve` images (scripts/Poker/trainGPU.py:152,120-131); the plotting service is outside
this engine's scope, `main()` passes no plotter, and the curves' data -- the per-episode reward and chip
sums -- are in the returned summary and in run_N.yaml instead.  `train_agent(_fused)` still take a `plotter` argument with the
reference's `plot_learning_curve` interface for a caller that brings one.

Keys this engine adds (all optional, defaults = the reference's behaviour where it has one):
  N_GPUS              1      one process per GPU; tables are sharded N_GAMES / N_GPUS per rank (sharding.py)
  SEED                0      Philox seed of the device shuffles, scripted draws and the learner's epsilon / dropout draws
  USE_PREFIXED_DECKS  false  decks from utils.performance.build_prefixed_deck_batch(seed = SEED + episode) instead of the device shuffle
  MAX_EPISODE_STEPS   null   cap on steps per episode (the reference's round-closing rule can livelock a table)
"""
from __future__ import annotations

from pathlib import Path

import yaml

PACKAGE_CONFIG_DIR = Path(__file__).resolve().parent.parent / "config"
POKER_ACTION_SPACE_N = 13                         # scripts/Poker/trainGPU.py:19

# what scripts/Poker/trainGPU.py:148-214 reads with config[...] (a missing one is a KeyError there, and here)
POKER_GPU_REQUIRED_KEYS = ("RESULTS_DIR", "ENV_ID", "AGENTS", "N_GAMES", "EPISODES", "STARTING_BBS", "NUM_PLAYERS", "STATE_SPACE",
                           "ACTION_SPACE", "W1", "W2", "K", "ALPHA", "UPDATE_FREQ", "GAMMA", "LEARNING_RATE", "WEIGHT_DECAY")
POKER_GPU_OPTIONAL_KEYS = ("PLOTTING", "BENCHMARKING")                       # config.get(...) in the reference (:152-153)
POKER_GPU_ENGINE_KEYS = {"N_GPUS": 1, "SEED": 0, "USE_PREFIXED_DECKS": False, "MAX_EPISODE_STEPS": None}


def get_config_file(file_name, config_dir: Path | None = None):
    """utils/config.py:6-15.  `file_name` is looked up under `config_dir` (default: this package's config/); an absolute
    or existing relative path is taken as it is (`--config /path/to/the/reference/config/pokerGPU.yaml`)."""
    path = Path(file_name)
    if not path.is_absolute() and not path.exists():
        path = (config_dir or PACKAGE_CONFIG_DIR) / file_name
    if not path.exists():
        return None
    with open(path, "r") as fh:
        return yaml.safe_load(fh)


def poker_gpu_arguments(config: dict) -> dict:
    """The arguments scripts/Poker/trainGPU.py:156-188 builds from the config, as plain dictionaries:
      load_gpu_agents(device, *agents_args), PokerQNetwork(weights_path, device, **q_network),
      gym.make(ENV_ID, device=, agents=, **env) == PokerGPU(device=, agents=, **env), train_agent(..., **train);
    plus `engine` (the keys this engine adds).  Values are handed over exactly as the YAML loader produced them."""
    missing = [k for k in POKER_GPU_REQUIRED_KEYS if k not in config]
    if missing:
        raise KeyError(missing[0])                                           # what config["..."] raises in the reference
    engine = {k: config.get(k, d) for k, d in POKER_GPU_ENGINE_KEYS.items()}
    return {
        "results_dir": config["RESULTS_DIR"],                                                                  # :150
        "env_id": config["ENV_ID"],
        "agents_args": (config["NUM_PLAYERS"], config["AGENTS"], config["STARTING_BBS"], POKER_ACTION_SPACE_N),  # :156-162
        "q_network": dict(gamma=config["GAMMA"], update_freq=config["UPDATE_FREQ"], state_dim=config["STATE_SPACE"],      # :163-172
                          action_dim=config["ACTION_SPACE"], learning_rate=config["LEARNING_RATE"], weight_decay=config["WEIGHT_DECAY"]),
        "env": dict(n_players=config["NUM_PLAYERS"] + 1, n_games=config["N_GAMES"], starting_bbs=config["STARTING_BBS"],     # :177-188
                    w1=config["W1"], w2=config["W2"], K=config["K"], alpha=config["ALPHA"]),
        "train": dict(episodes=config["EPISODES"], n_games=config["N_GAMES"]),                                      # :192-201
        "plotting": config.get("PLOTTING"), "benchmarking": config.get("BENCHMARKING"),                          # :152-153
        "engine": engine,
    }
