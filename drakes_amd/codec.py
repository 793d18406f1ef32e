"""The formats either side of the sampling path: DNA token <-> string codec
(reference /root/reference/drakes_dna/dataloader_gosai.py:11-37) and checkpoint loading
(finetune_reward_bp.py:117,174; eval.ipynb: Lightning `.ckpt` files and bare state_dicts).  Host-side byte work:
numpy, no kernels.
"""
import numpy as np
import torch

DNA_ALPHABET = {'A': 0, 'C': 1, 'G': 2, 'T': 3}          # dataloader_gosai.py:11 (the mask token, 4, has no letter)
INDEX_TO_DNA = {v: k for k, v in DNA_ALPHABET.items()}
_LETTERS = np.frombuffer(b'ACGT', dtype=np.uint8)
_CODES = np.full(256, 255, dtype=np.uint8)
for _c, _i in DNA_ALPHABET.items():
    _CODES[ord(_c)] = _i


def dna_detokenize(seq):
    """dataloader_gosai.py:16-17.  seq: iterable of ints in 0..3."""
    return ''.join(INDEX_TO_DNA[int(i)] for i in seq)


def batch_dna_detokenize(batch_seq):
    """dataloader_gosai.py:19-26.  [batch, seq_len] ints (numpy or CPU/GPU tensor) -> list of strings."""
    if torch.is_tensor(batch_seq):
        batch_seq = batch_seq.detach().cpu().numpy()
    a = np.asarray(batch_seq)
    if a.size and (a.min() < 0 or a.max() > 3):
        raise IndexError('token outside 0..3 (the mask token has no letter)')      # the reference's lookup raises too
    letters = _LETTERS[a.astype(np.int64)]
    return [row.tobytes().decode('ascii') for row in letters.reshape(-1, a.shape[-1])] if a.ndim > 1 else \
        [letters.tobytes().decode('ascii')]


def dna_tokenize(seq):
    """dataloader_gosai.py:28-29.  Unknown letters raise KeyError like the reference's dict lookup."""
    return [DNA_ALPHABET[c] for c in seq]


def batch_dna_tokenize(batch_seq):
    """dataloader_gosai.py:31-37.  list of equal-length strings -> int64 numpy array [batch, seq_len]."""
    if len(batch_seq) == 0:
        return np.zeros((0, 0), dtype=np.int64)
    n = len(batch_seq[0])
    if any(len(s) != n for s in batch_seq):
        raise ValueError('sequences of different length (np.array of ragged lists in the reference)')
    raw = np.frombuffer(''.join(batch_seq).encode('ascii'), dtype=np.uint8)
    codes = _CODES[raw]
    if (codes == 255).any():
        bad = chr(int(raw[np.argmax(codes == 255)]))
        raise KeyError(bad)
    return codes.reshape(len(batch_seq), n).astype(np.int64)


def load_checkpoint_state_dict(path, map_location='cpu'):
    """Weights of a DRAKES DNA model from either on-disk form, as a dict with the reference keys `backbone.*`:
      * a Lightning checkpoint (`Diffusion.load_from_checkpoint`, finetune_reward_bp.py:174): dict with 'state_dict';
      * a bare state_dict (`torch.save(new_model.state_dict(), ...)`, finetune_reward_bp.py:117).
    Loaded with weights_only=True (nothing in the file is executed).  Non-backbone entries (EMA shadow, metrics) are
    dropped."""
    obj = torch.load(path, map_location=map_location, weights_only=True)
    sd = obj['state_dict'] if isinstance(obj, dict) and 'state_dict' in obj else obj
    if not isinstance(sd, dict):
        raise ValueError(f'{path}: not a state_dict or a Lightning checkpoint')
    out = {k: v for k, v in sd.items() if k.startswith('backbone.') and torch.is_tensor(v)}
    if not out:
        raise ValueError(f'{path}: no backbone.* tensors found')
    return out
