"""``python -m llark_amd.clap.embed_cli --input-dir D --output-dir O --ckpt-file F``: the reference's
``scripts/clap/clap_embeddings.py`` job (:170-253) without Beam -- every ``*.wav`` under ``--input-dir`` becomes
``<output-dir>/<name>.npy`` holding its ``(1, 512)`` CLAP embedding.  Where the reference fans files out to Dataflow workers,
this shards the sorted file list over the launched ranks (one process per GPU, RANK / WORLD_SIZE from the launcher, no
collective: clips are independent) and embeds ``--batch-size`` clips per pass on the GPU.

Audio reading: RIFF wav via scipy (int16 / int32 / float), mixed down to mono, resampled to 48 kHz when the file's rate
differs with a polyphase filter to libsoxr's HQ specification (the reference calls ``librosa.resample`` of an unpinned librosa
inside ``read_wav``, m2t/gcs_utils.py:133-135 = soxr_hq: same pass band / rejection, not the same taps -- feed 48 kHz files for
exact agreement)."""
from __future__ import annotations

import argparse
import os
from typing import Iterator, List, Optional, Tuple

import numpy as np

from .frontend import SAMPLE_RATE


def list_wavs(input_dir: str) -> List[str]:
    out = []
    for root, _, files in os.walk(input_dir):
        out += [os.path.join(root, f) for f in files if f.lower().endswith(".wav")]
    return sorted(out)


def shard(paths: List[str], rank: int, world: int) -> List[str]:
    return paths[rank::world]


def read_wav_48k(path: str) -> np.ndarray:
    from scipy.io import wavfile

    from ..jukebox.resample import resample

    sr, x = wavfile.read(path)
    if x.dtype == np.int16:
        x = x.astype(np.float32) / 32768.0
    elif x.dtype == np.int32:
        x = x.astype(np.float32) / 2147483648.0
    elif x.dtype == np.uint8:
        x = (x.astype(np.float32) - 128.0) / 128.0
    else:
        x = x.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1)
    if sr != SAMPLE_RATE:
        # m2t/gcs_utils.py:133-135: librosa.resample(samples, orig_sr, target_sr) of an unpinned (current) librosa = soxr_hq
        x = resample(x, int(sr), SAMPLE_RATE, "soxr_hq")
    return x


def iter_batches(paths: List[str], batch_size: int) -> Iterator[Tuple[List[str], List[np.ndarray]]]:
    """Skips unreadable / empty files with a message (the reference filters elements whose features are None)."""
    names, clips = [], []
    for p in paths:
        try:
            x = read_wav_48k(p)
            if x.size == 0:
                raise ValueError("no samples")
        except Exception as e:  # noqa: BLE001 -- any decode failure drops the file, like the reference's filter step
            print(f"[WARN] skipping {p}: {e}")
            continue
        names.append(p)
        clips.append(x)
        if len(clips) == batch_size:
            yield names, clips
            names, clips = [], []
    if clips:
        yield names, clips


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--input-dir", required=True, help="path to directory containing wav audio.")
    ap.add_argument("--output-dir", required=True, help="directory to output files.")
    ap.add_argument("--ckpt-file", required=True, help="Path to a CLAP checkpoint file (should end with .pt).")
    ap.add_argument("--batch-size", type=int, default=64, help="clips per GPU pass")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--seed", type=int, default=0, help="seed of the rand_trunc crop offsets")
    args = ap.parse_args(argv)

    import torch

    from .module import HipClapModule

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    model = HipClapModule(enable_fusion=False, amodel="HTSAT-base", device=device, precision=args.precision)
    model.load_ckpt(args.ckpt_file)
    os.makedirs(args.output_dir, exist_ok=True)
    mine = shard(list_wavs(args.input_dir), rank, world)
    print(f"[INFO] rank {rank}/{world}: processing {len(mine)} files")
    rng = np.random.default_rng(args.seed + rank)
    done = 0
    for names, clips in iter_batches(mine, args.batch_size):
        emb = model.get_audio_embedding_from_data(clips, rng=rng)
        for p, e in zip(names, emb):
            np.save(os.path.join(args.output_dir, os.path.basename(p)[:-4] + ".npy"), e[None].astype(np.float32))
        done += len(names)
    print(f"[INFO] rank {rank}: wrote {done} embeddings to {args.output_dir}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
