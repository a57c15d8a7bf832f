"""Command-line decode drivers — the caller contract of the reference's `decode.py`,
`decode_tweedie.py`, `decode_TDS.py`, `decode_DPS.py` (reference decode.py:52-211): seed, build the
nets, run `BaseModel.controlled_decode*`, write `./log/{task}-{reward_name}[_tw|_TDS|_DPS].npz` with
the two arrays `decoding` and `baseline` (decode.py:117, decode_tweedie.py:118, decode_TDS.py:118,
decode_DPS.py:119).

Offline there are no W&B artifacts, so nets are random-init unless state_dicts are given
(`--diffusion_ckpt`, `--load_checkpoint_path`, `--reward_ckpt`; reference key names load unchanged).
Only the flags that reach the decode path are kept; the reference's ~40 vestigial training flags are not."""
import argparse
import os
import random

import numpy as np
import torch

SUFFIX = {"mc": "", "tweedie": "_tw", "tds": "_TDS", "dps": "_DPS"}


def set_seed(seed):
    """reference decode.py:31-35"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def build_parser(method="mc"):
    p = argparse.ArgumentParser(description=f"SVDD decode ({method}) on MI355X")
    p.add_argument("--task", default="dna", choices=["dna", "rna"])                 # decode.py:130
    p.add_argument("--reward_name", default=None, help="HepG2 (dna) / MRL (rna) by default")
    p.add_argument("--batch_size", type=int, default=256)                            # :163
    p.add_argument("--sample_M", type=int, default=10)                               # :165
    p.add_argument("--val_batch_num", type=int, default=1)                           # :167
    p.add_argument("--seed", type=int, default=44)                                   # :181
    p.add_argument("--method", default=method, choices=list(SUFFIX))
    p.add_argument("--tweedie", default="True", help='"True": posterior-mean scoring (decode_tweedie.py:206)')
    p.add_argument("--alpha", type=float, default=0.5)                               # decode_TDS.py:183
    p.add_argument("--guidance_scale", type=float, default=1e5)                      # decode_DPS.py:184
    p.add_argument("--model", default="convgru", choices=["convgru", "enformer"],
                   help="value-function trunk: convgru = ConvGRUTrunk + ConvHead (Enformer.py:32-49; BASELINE configs 1-3, SURVEY "
                        "section 8d); enformer = the 230 M-parameter EnformerTrunk(7, 1536, 11, 8, 64) + ConvHead(1, 3072) the "
                        "reference's decode.py:78-80 builds for --model enformer (BASELINE configs[3])")                   # decode.py:149
    p.add_argument("--precision", default="f32", choices=["f32", "f16x3", "bf16x3", "f16", "bf16"],
                   help="f32: exact fp32 net kernels (default, the parity reference); x3: fp32 operands split hi + lo on the 16-bit "
                        "matrix cores (f16x3: fp32-class error; bf16x3: a 16-bit operand, 1e-5-class); f16 / bf16: one pass. With --model enformer any non-f32 mode runs the "
                        "hand-written trunk kernels (bf16x3 / bf16); f32 runs the PyTorch module")
    p.add_argument("--rng", default="replay", choices=["replay", "philox"],
                   help="replay: the reference's torch-CPU RNG stream; philox: in-kernel counter RNG")
    p.add_argument("--diffusion_ckpt", default=None)
    p.add_argument("--load_checkpoint_path", default=None, help="value-function checkpoint ('model_state_dict')")
    p.add_argument("--reward_ckpt", default=None)
    p.add_argument("--out_dir", default="./log")
    p.add_argument("--presample", action="store_true",
                   help="pre-sample val_batch_num batches at construction like the reference's BaseModel.__init__")
    return p


def run(args):
    from . import synthetic
    from .harness import BaseModel
    from .value_nets import load_reference_state_dict

    set_seed(args.seed)
    reward_name = args.reward_name or ("HepG2" if args.task == "dna" else "MRL")
    ref_model, embedding, head, reward = synthetic.build(args.task, "cuda", seed=args.seed, value=args.model)
    ref_model.precision = args.precision
    if args.diffusion_ckpt:
        sd = torch.load(args.diffusion_ckpt, map_location="cpu")
        ref_model.load_state_dict(sd.get("state_dict", sd), strict=False)
    if args.load_checkpoint_path:
        sd = torch.load(args.load_checkpoint_path, map_location="cpu")["model_state_dict"]
        emb_sd = {k[len("embedding."):]: v for k, v in sd.items() if k.startswith("embedding.")}
        if args.model == "enformer":                       # the reference's wrapper names -> this module's (enformer_value.py)
            from .enformer_value import load_reference_state_dict as load_trunk
            load_trunk(embedding, emb_sd)
        else:
            load_reference_state_dict(embedding, emb_sd)
        load_reference_state_dict(head, {k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")})
    if args.reward_ckpt:
        sd = torch.load(args.reward_ckpt, map_location="cpu")
        load_reference_state_dict(reward, sd.get("model_state_dict", sd))
    ref_model.rng_mode, ref_model.philox_seed = args.rng, args.seed
    model = BaseModel(embedding, head, ref_model, reward, args.batch_size, task=args.task,
                      val_batch_num=args.val_batch_num if args.presample else 0).cuda().eval()
    if args.method == "mc":
        out = model.controlled_decode(gen_batch_num=args.val_batch_num, sample_M=args.sample_M)
    elif args.method == "tweedie":
        out = model.controlled_decode_tweedie(gen_batch_num=args.val_batch_num, sample_M=args.sample_M, options=args.tweedie)
    elif args.method == "tds":
        out = model.controlled_decode_TDS(gen_batch_num=args.val_batch_num, sample_M=args.sample_M, alpha=args.alpha)
    else:
        out = model.controlled_decode_DPS(gen_batch_num=args.val_batch_num, sample_M=args.sample_M,
                                          guidance_scale=args.guidance_scale)
    gen_samples, value_func_preds, reward_model_preds, selected_baseline_preds, baseline_preds = out
    os.makedirs(args.out_dir, exist_ok=True)
    path = os.path.join(args.out_dir, f"{args.task}-{reward_name}{SUFFIX[args.method]}")
    np.savez(path, decoding=reward_model_preds.cpu().numpy(), baseline=baseline_preds.cpu().numpy())
    print(f"wrote {path}.npz: decoding mean {reward_model_preds.mean().item():.4f} (n={reward_model_preds.numel()}), "
          f"baseline mean {baseline_preds.mean().item():.4f}")
    return path + ".npz", out


def main(method="mc", argv=None):
    return run(build_parser(method).parse_args(argv))


if __name__ == "__main__":
    main()
