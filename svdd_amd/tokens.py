"""Token <-> nucleotide helpers of the reference's data layer that the decode outputs pass through
(reference `dataloader_gosai.py:13-32` `DNA_ALPHABET`, `dna_detokenize`, `batch_dna_detokenize`, and the
`DNASequenceDetokenizer` class `:35-70`): 0..3 = A, C, G, T; 4 = MASK (never present in a finished decode).
"""
import numpy as np
import torch

DNA_ALPHABET = {"A": 0, "C": 1, "G": 2, "T": 3}
INDEX_TO_DNA = {v: k for k, v in DNA_ALPHABET.items()}
MASK_INDEX = 4
_LUT = np.frombuffer(b"ACGTN", dtype="S1")          # index 4 (MASK) and anything above print as N


def _as_index_array(batch_seq):
    if isinstance(batch_seq, torch.Tensor):
        batch_seq = batch_seq.detach().cpu().numpy()
    return np.minimum(np.asarray(batch_seq).astype(np.int64), 4)


def dna_detokenize(seq):
    """One sequence of token ids -> string."""
    return _LUT[_as_index_array(seq)].tobytes().decode("ascii")


def batch_dna_detokenize(batch_seq):
    """[batch, length] token ids (numpy or tensor) -> list of strings."""
    idx = _as_index_array(batch_seq)
    return [row.tobytes().decode("ascii") for row in _LUT[idx.reshape(-1, idx.shape[-1])]]


def dna_tokenize(seq):
    """String -> int64 numpy array of token ids (unknown letters -> MASK)."""
    return np.array([DNA_ALPHABET.get(ch, MASK_INDEX) for ch in seq.upper()], dtype=np.int64)


class DNASequenceDetokenizer:
    """Object form used by the reference's evaluation code: `.detokenize(batch) -> list[str]`."""

    dna_alphabet = DNA_ALPHABET
    index_to_dna = INDEX_TO_DNA
    unknown_char = "N"

    def detokenize(self, batch_seq):
        return batch_dna_detokenize(batch_seq)
