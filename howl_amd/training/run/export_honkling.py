"""``python -m training.run.export_honkling -i model-best.pt.bin -o weights.js --name RES8``: the wire format of
``training/run/export_honkling.py:8-35`` (a JavaScript assignment of the ``state_dict`` as nested lists, plus the three
unit ``scale{1,3,5}.scale`` vectors Honkling's RES8 expects).  Checkpoints written by ``howl_amd`` use the reference's
``state_dict`` keys, so the export is key-for-key what stock Howl produces."""
import argparse
import json

import torch


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--input-file", "-i", type=str, required=True)
    ap.add_argument("--output-file", "-o", type=str, required=True)
    ap.add_argument("--name", type=str, required=True)
    args = ap.parse_args(argv)
    state_dict = torch.load(args.input_file, map_location="cpu")
    json_dict = {}
    if args.name == "RES8":
        for i in (1, 3, 5):
            state_dict[f"scale{i}.scale"] = torch.ones(45)
    for key, tensor in state_dict.items():
        json_dict[key] = tensor.tolist()
    with open(args.output_file, "w") as f:
        f.write(f"weights['{args.name}'] = ")
        json.dump(json_dict, f)


if __name__ == "__main__":
    main()
