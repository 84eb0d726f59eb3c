"""Fixture for the text encoder, produced by the installed transformers' own T5EncoderModel --
the third-party module the reference calls at models/audio_distilled_model.py:97-98,208-214 (pinned there to
transformers==4.29.2; the encoder arithmetic is unchanged in the installed release) -- with the build's deterministic
weights (FLAN-T5 checkpoints are not available offline):

    python tests/golden/make_golden_t5.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402


def run(cfg, B, L, tag):
    from transformers import T5Config, T5EncoderModel
    m = T5EncoderModel(T5Config(**cfg)).eval()
    sd = cases.t5_weights(cfg)
    assert list(m.state_dict().keys()) == list(__import__("consistencytta_amd.spec", fromlist=["x"]).t5_encoder_param_spec(cfg).keys())
    m.load_state_dict(sd)
    ids, mask = cases.t5_inputs(cfg, B, L, tag)
    with torch.no_grad():
        out = m(input_ids=ids, attention_mask=mask)[0]
    return out.numpy()


def main():
    import transformers
    out = {"tiny": run(cases.TINY_T5, 3, 13, "t5_tiny"), "tiny_long": run(cases.TINY_T5, 2, 150, "t5_long"),
           "wide": run(cases.WIDE_T5, 2, 16, "t5_wide"), "transformers_version": np.array(transformers.__version__)}
    path = os.path.join(HERE, "t5_encoder.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    main()
