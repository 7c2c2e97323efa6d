"""The entry point consumes the reference's config/pokerGPU.yaml as it is: flat keys
(utils/config.py:6-15 loader; scripts/Poker/trainGPU.py:148-214 consumer), values passed through untouched."""
from pathlib import Path

import pytest
import yaml

from pulselib_amd.utils import config as cfg

# The reference's config/pokerGPU.yaml:1-37, re-typed key for key (a config schema is data, not code)
REFERENCE_POKER_GPU_YAML = """
RESULTS_DIR: "PokerGPU"
ENV_ID: "Pulse-Poker-GPU-v1"
PLOT_FILENAME: ""
SCORES_FILENAME: ""
AGENTS:
  - "tight_aggressive"
  - "heuristic_hands"
  - "heuristic_hands"
  - "loose_passive"
  - "tight_aggressive"
  - "random"
  - "loose_passive"
  - "small_ball"
  - "tight_aggressive"

CAPACITY: 100000
N_GAMES: 2000000
EPISODES: 1000

STARTING_BBS: 100
NUM_PLAYERS: 9

STATE_SPACE: 40
ACTION_SPACE: 13

W1: .5
W2: .3
W1_DECAY: .99999
W2_DECAY: .9999
K: 100
UPDATE_FREQ: 20
GAMMA: .95
ALPHA: 50

LEARNING_RATE: 2e-4
WEIGHT_DECAY: 1e-5
"""


def _reference_constructor_arguments(config):
    """What scripts/Poker/trainGPU.py:156-201 passes on, written out from its lines (the expectation of this test)."""
    return {
        "results_dir": config["RESULTS_DIR"],                                                         # :150
        "env_id": config["ENV_ID"],                                                                   # :178
        "agents_args": (config["NUM_PLAYERS"], config["AGENTS"], config["STARTING_BBS"], 13),         # :156-162 (POKER_ACTION_SPACE_N = 13, :19)
        "q_network": {"gamma": config["GAMMA"], "update_freq": config["UPDATE_FREQ"], "state_dim": config["STATE_SPACE"],   # :163-172
                      "action_dim": config["ACTION_SPACE"], "learning_rate": config["LEARNING_RATE"], "weight_decay": config["WEIGHT_DECAY"]},
        "env": {"n_players": config["NUM_PLAYERS"] + 1, "n_games": config["N_GAMES"], "starting_bbs": config["STARTING_BBS"],   # :177-188
                "w1": config["W1"], "w2": config["W2"], "K": config["K"], "alpha": config["ALPHA"]},
        "train": {"episodes": config["EPISODES"], "n_games": config["N_GAMES"]},                      # :192-201
        "plotting": config.get("PLOTTING"), "benchmarking": config.get("BENCHMARKING"),              # :152-153
    }


def test_reference_config_file_goes_through_the_loader_unchanged(tmp_path):
    path = tmp_path / "pokerGPU.yaml"
    path.write_text(REFERENCE_POKER_GPU_YAML)
    config = cfg.get_config_file(str(path))
    assert config == yaml.safe_load(REFERENCE_POKER_GPU_YAML)
    assert config["LEARNING_RATE"] == "2e-4" and config["WEIGHT_DECAY"] == "1e-5"      # PyYAML reads these as strings; the learner floats them (Player.py:224-225)
    got = cfg.poker_gpu_arguments(config)
    want = _reference_constructor_arguments(config)
    assert {k: got[k] for k in want} == want
    assert got["env"]["n_players"] == 10 and got["env"]["n_games"] == 2000000 and got["q_network"]["state_dim"] == 40
    assert got["engine"] == {"N_GPUS": 1, "SEED": 0, "USE_PREFIXED_DECKS": False, "MAX_EPISODE_STEPS": None}     # defaults: the reference's behaviour


def test_shipped_config_is_the_reference_schema_plus_engine_keys():
    shipped = cfg.get_config_file("pokerGPU.yaml")
    reference = yaml.safe_load(REFERENCE_POKER_GPU_YAML)
    assert {k: shipped[k] for k in reference} == reference                      # same keys, same values
    assert set(shipped) - set(reference) == set(cfg.POKER_GPU_ENGINE_KEYS)
    a = cfg.poker_gpu_arguments(shipped)
    assert a["engine"]["N_GPUS"] == 1 and a["engine"]["USE_PREFIXED_DECKS"] is False and a["engine"]["SEED"] == 20260401


def test_loader_contracts():
    assert cfg.get_config_file("no_such_file.yaml") is None                     # utils/config.py:9-10
    config = yaml.safe_load(REFERENCE_POKER_GPU_YAML)
    del config["UPDATE_FREQ"]
    with pytest.raises(KeyError, match="UPDATE_FREQ"):                          # config["UPDATE_FREQ"] in the reference
        cfg.poker_gpu_arguments(config)
    config = yaml.safe_load(REFERENCE_POKER_GPU_YAML)
    config["BENCHMARKING"] = {"enabled": False, "mask": {"training_summary": False}}
    from pulselib_amd.utils.benchmarking import YamlBenchmarker
    b = YamlBenchmarker.from_config(cfg.poker_gpu_arguments(config)["benchmarking"])
    assert b.enabled is False and b.feature_mask["training_summary"] is False


def test_learner_accepts_the_yaml_strings():
    """LEARNING_RATE / WEIGHT_DECAY arrive as the strings PyYAML makes of `2e-4` / `1e-5`."""
    import torch
    from pulselib_amd.environments.Poker.qnetwork import PokerQNetwork
    config = yaml.safe_load(REFERENCE_POKER_GPU_YAML)
    a = cfg.poker_gpu_arguments(config)
    q = PokerQNetwork(None, torch.device("cpu"), **a["q_network"])
    assert q.lr == 2e-4 and q.wd == 1e-5 and q.update_freq == 20
