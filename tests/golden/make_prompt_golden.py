"""Golden vectors for the host-side prompt glue, produced by the REAL reference functions
(m2t/data_modules.py, m2t/conversation_utils.py, m2t/generate.py) driven by tests/toy_tokenizer.py.
Run in the build container only:  python tests/golden/make_prompt_golden.py"""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
for name in ("braceexpand", "msgspec", "webdataset"):         # not installed; only imported at module level
    m = types.ModuleType(name)
    m.WebDataset = object
    sys.modules[name] = m

import numpy as np  # noqa: E402
import torch  # noqa: E402
from m2t import data_modules as DM  # noqa: E402  (the reference)
from m2t.conversation_utils import extract_prompt_tokens, extract_response_tokens  # noqa: E402
from m2t.generate import KeywordsStoppingCriteria  # noqa: E402
from toy_tokenizer import ToyTokenizer  # noqa: E402


def main():
    tok = ToyTokenizer()
    tok.add_tokens(["<audio_patch>"], special_tokens=True)
    tok.add_tokens(["<audio_start>", "<audio_end>"], special_tokens=True)
    cfg = dict(is_multimodal=True, sep_audio_conv_front=False, use_audio_start_end=True)
    out = {"header": DM.DEFAULT_CONVERSATION_HEADER}
    cases = []
    for i, (prompt, frames, audio_first) in enumerate([("Describe the tempo of this clip .", 5, True),
                                                        ("What instruments are playing ?", 3, False)]):
        text = DM.concat_audio_token_and_prompt(prompt, audio_first)
        enc = np.arange(frames * 4, dtype=np.float32).reshape(frames, 4)
        elem = {"audio_encoding": enc, "audio_encoding_shape": list(enc.shape), "example_id": f"ex{i}", "id": f"ex{i}",
                "conversations": [{"from": "human", "value": text}, {"from": "gpt", "value": "a slow ballad" if i else "<empty>"}]}
        elem = DM.preprocess_multimodal_mappable(elem, cfg)
        conv_text = [dict(c) for c in elem["conversations"]]
        res = DM.preprocess_for_lm_mappable(elem, tokenizer=tok)
        end_seq = tok("\n### Assistant:").input_ids[1:]        # drop the BOS like get_prompt_end_token_sequence does
        prompt_ids = extract_prompt_tokens(res["input_ids"], end_seq)
        resp_ids = extract_response_tokens(res["input_ids"], end_seq)
        cases.append(dict(prompt=prompt, frames=frames, audio_first=audio_first, text=text, conversations=conv_text,
                          input_ids=res["input_ids"].tolist(), labels=res["labels"].tolist(), end_seq=end_seq,
                          prompt_ids=prompt_ids.tolist(), response_ids=resp_ids.tolist(),
                          audio_shape=list(res["audio_encoding"].shape)))
    out["cases"] = cases
    # collator on the two instances (ragged lengths, different frame counts -> list of encodings)
    inst = [dict(input_ids=torch.tensor(c["input_ids"]), labels=torch.tensor(c["labels"]),
                 audio_encoding=torch.zeros(c["frames"], 4)) for c in cases]
    batch = DM.DataCollatorForSupervisedDataset(tokenizer=tok)(inst)
    out["collated"] = dict(input_ids=batch["input_ids"].tolist(), labels=batch["labels"].tolist(),
                           attention_mask=batch["attention_mask"].long().tolist(),
                           encodings_is_list=isinstance(batch["audio_encodings"], list))
    inst2 = [dict(input_ids=torch.tensor(cases[0]["input_ids"]), labels=torch.tensor(cases[0]["labels"]),
                  audio_encoding=torch.zeros(5, 4)) for _ in range(2)]
    batch2 = DM.DataCollatorForSupervisedDataset(tokenizer=tok)(inst2)
    out["collated_equal_shapes_is_tensor"] = bool(torch.is_tensor(batch2["audio_encodings"]))
    # stopping criterion
    ids0 = torch.tensor([cases[0]["prompt_ids"]])
    crit = KeywordsStoppingCriteria(keywords=["###"], tokenizer=tok, input_ids=ids0)
    hash_id = tok("###").input_ids[1]
    seqs = {"no_stop": cases[0]["prompt_ids"] + [tok._id("slow"), tok._id("ballad")],
            "stop_last": cases[0]["prompt_ids"] + [tok._id("slow"), hash_id]}
    out["stopping"] = {k: bool(crit(torch.tensor([v]), None)) for k, v in seqs.items()}
    out["stopping_seqs"] = seqs
    out["vocab"] = tok.vocab
    json.dump(out, open(os.path.join(HERE, "prompt_glue.json"), "w"), indent=1)
    print("wrote prompt_glue.json", out["stopping"], [len(c["input_ids"]) for c in cases])


if __name__ == "__main__":
    main()
