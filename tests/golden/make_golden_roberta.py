"""Fixture for the CLAP text tower: `transformers.RobertaModel`, the module the reference's
`clap_module/model.py:504,633-641` instantiates (`RobertaModel.from_pretrained('roberta-base')`, read through
`["pooler_output"]`).  The roberta-base checkpoint is not available offline, so -- as for FLAN-T5 -- the architecture is
pinned against the installed transformers' own module with the build's deterministic weights:

    python tests/golden/make_golden_roberta.py

Cases: a small model (hidden 128, 2 layers) and roberta-base's widths at 1 layer; right-padded batches (pad id 1) so
that the position ids (cumulative count of non-pad tokens + 1) and the key mask are exercised."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
from consistencytta_amd import spec  # noqa: E402


def main():
    from transformers import RobertaConfig, RobertaModel
    out = {}
    for tag, cfg in (("tiny", cases.TINY_ROBERTA), ("wide", cases.WIDE_ROBERTA)):
        hf = RobertaModel(RobertaConfig(**cfg, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)).eval()
        sd = cases.roberta_weights(cfg)
        missing = hf.load_state_dict(sd, strict=False)
        assert not [k for k in missing.missing_keys if "position_ids" not in k and "token_type_ids" not in k], missing
        assert not missing.unexpected_keys, missing
        # same parameter set; the ORDER of the embedding tables differs between transformers releases (the spec follows
        # 4.29.2, the reference's pin: word, position, token_type, LayerNorm)
        assert sorted(k for k in hf.state_dict() if not k.endswith(("position_ids", "token_type_ids"))) == sorted(sd)
        ids, mask = cases.roberta_inputs(cfg, 3, 20, tag)
        with torch.no_grad():
            o = hf(input_ids=ids, attention_mask=mask)
        out[tag + "_pooler"] = o["pooler_output"].numpy()
        out[tag + "_last"] = o["last_hidden_state"].numpy()
        print(tag, float(o["pooler_output"].abs().max()), float(o["last_hidden_state"].abs().max()))
    path = os.path.join(HERE, "roberta.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
