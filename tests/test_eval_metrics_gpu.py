"""-m gpu: the run bookkeeping around the hot path -- metrics.jsonl (the reference's TensorBoard scalars,
paac.py:130-135,176-180) and the evaluation harness (test.py:50-88) restoring a checkpoint the learner wrote."""
import json
import os
import tempfile

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_train_writes_metrics_then_eval_restores_checkpoint(capsys):
    from paac_amd import logger_utils, train
    from paac_amd import test as harness
    from paac_amd.paac import PAACLearner
    folder = tempfile.mkdtemp(prefix="paac_eval_")
    args = train.get_arg_parser().parse_args(["-g", "breakout", "--arch", "NIPS", "-ec", "32", "-ew", "0",
                                              "--max_global_steps", str(32 * 5 * 130), "-df", folder,
                                              "--synthetic_terminal_p", "0.02", "--sampler", "numpy"])
    logger_utils.save_args(args, folder)
    network_creator, env_creator = train.get_network_and_environment_creator(args)
    learner = PAACLearner(network_creator, env_creator, args)
    np.random.seed(3)
    learner.train()                       # device loop; cleanup() saves network + optimizer checkpoints
    want = learner.network.get_parameters()

    recs = [json.loads(l) for l in open(os.path.join(folder, "metrics.jsonl"))]
    progress = [r for r in recs if r["kind"] == "progress"]
    episodes = [r for r in recs if r["kind"] == "episode"]
    assert len(progress) == 2             # every 2048/32 = 64 cycles (paac.py:172), 130 cycles run
    assert progress[0]["global_step"] == 64 * 160 and progress[1]["global_step"] == 128 * 160
    for r in progress:
        assert r["steps_per_s"] > 0 and np.isfinite([r["loss"], r["actor_loss"], r["critic_loss"], r["entropy"]]).all()
        assert 0 < r["lr"] < 0.0224 and r["grad_norm"] > 0 and 0 < r["entropy"] <= np.log(4) + 1e-5
    assert len(episodes) > 50 and all(e["length"] >= 1 for e in episodes)   # p = 0.02 per step, 20k steps

    rewards = harness.main(["-f", folder, "-tc", "3", "-np", "5"])
    out = capsys.readouterr().out
    assert "Performed 3 tests for breakout." in out and "Mean:" in out and "Std:" in out
    assert rewards.shape == (3,) and np.isfinite(rewards).all()

    # the harness restored exactly what the learner saved
    from paac_amd.session import Saver, checkpoint_key
    path = Saver.latest_checkpoint(os.path.join(folder, "checkpoints"))
    assert path.endswith("-%d.npz" % (130 * 160))
    with np.load(path) as z:
        assert "local_learning_1/conv1_weights" in z.files and "local_learning_2/actor_output_biases" in z.files
        for k, v in want.items():
            assert np.array_equal(z[checkpoint_key("local_learning", k)], v)
    opt = Saver.latest_checkpoint(os.path.join(folder, "optimizer_checkpoints"))
    with np.load(opt) as z:       # both RMSProp slots of every variable, under the reference's names
        assert len(z.files) == 2 * len(want)
        assert "local_learning_1/fc3_weights/OptimizerVariables" in z.files
        assert "local_learning_2/critic_output_weights/OptimizerVariables_1" in z.files
