"""Oracle (test infrastructure, numpy): beam search bookkeeping as the reference runs it.

Restates the loop of DiCoWGenerationMixin._beam_search (reference src/models/dicow/generation.py:815-1153) and the helpers it
inherits from transformers' GenerationMixin (third-party, pinned 4.55 in the reference; same functions in the installed 5.x):
top-2K continuations over beams x vocabulary, the K best unfinished ones keep running, finished ones (eos, or the length
limit) compete for the K result slots with length-penalised scores, the early-stop heuristic, the final selection.
Pinned against tests/golden/f15_beam_search.npz (the transformers helpers driven in the reference's order).
Only tests/ may import this module.
"""
import numpy as np

NEG = np.float32(-1.0e9)
f32 = np.float32


def _topk(x, k):
    """torch.topk along the last axis (descending, first index wins ties)."""
    idx = np.argsort(-x, axis=-1, kind="stable")[..., :k]
    return np.take_along_axis(x, idx, -1), idx


def _gather(t, idx):
    while idx.ndim < t.ndim:
        idx = idx[..., None]
    return np.take_along_axis(t, idx, 1)


class BeamState:
    def __init__(self, prompt, K, V, max_length, eos, length_penalty=1.0, early_stopping=False):
        B, P = prompt.shape
        self.B, self.K, self.V, self.P, self.max_length, self.eos = B, K, V, P, max_length, eos
        self.lp, self.es = float(length_penalty), early_stopping
        self.cur_len = P
        self.running_sequences = np.full((B, K, max_length), eos, dtype=np.int64)
        self.running_sequences[:, :, :P] = prompt[:, None, :]
        self.sequences = self.running_sequences.copy()
        self.running_beam_scores = np.zeros((B, K), dtype=f32)
        self.running_beam_scores[:, 1:] = NEG
        self.beam_scores = np.full((B, K), NEG, dtype=f32)
        self.is_sent_finished = np.zeros((B, K), dtype=bool)
        self.unsat = np.ones((B, 1), dtype=bool)
        self.running_beam_indices = np.full((B, K, max_length - P), -1, dtype=np.int32)
        self.beam_indices = self.running_beam_indices.copy()
        self.done = False

    def flat_sequences(self):
        return self.running_sequences[:, :, :self.cur_len].reshape(self.B * self.K, self.cur_len)

    def step(self, log_probs):
        """log_probs fp32 [B*K, V]: processed next-token log-probabilities of the running beams.  Returns beam_idx [B*K]:
        which previous beam each new running beam continues (for the KV caches and the CTC rescorer)."""
        B, K, V, P, cur = self.B, self.K, self.V, self.P, self.cur_len
        acc = (log_probs.reshape(B, K, V).astype(f32) + self.running_beam_scores[:, :, None]).reshape(B, K * V)
        tk_lp, tk = _topk(acc, 2 * K)
        src = tk // V
        tk_seq = _gather(self.running_sequences, src).copy()
        tk_idx = _gather(self.running_beam_indices, src).copy()
        tk_seq[:, :, cur] = tk % V
        tk_idx[:, :, cur - P] = src + np.arange(B)[:, None] * K
        hits = (tk_seq[:, :, cur] == self.eos) | (cur + 1 >= self.max_length)
        # running beams: the K best continuations that did not stop
        run_lp = (tk_lp + hits.astype(f32) * NEG).astype(f32)
        _, nxt = _topk(run_lp, K)
        self.running_sequences = _gather(tk_seq, nxt)
        self.running_beam_scores = _gather(run_lp, nxt)
        self.running_beam_indices = _gather(tk_idx, nxt)
        # finished beams: only a stop among the top K counts; length-penalised score
        just = hits & (np.arange(2 * K) < K)[None, :]
        fin = (tk_lp / f32((cur + 1 - P) ** self.lp)).astype(f32)
        full = self.is_sent_finished.all(axis=-1, keepdims=True) & (self.es is True)
        fin = fin + full.astype(f32) * NEG
        fin = fin + (~self.unsat).astype(f32) * NEG
        fin = fin + (~just).astype(f32) * NEG
        m_seq = np.concatenate([self.sequences, tk_seq], 1)
        m_sc = np.concatenate([self.beam_scores, fin.astype(f32)], 1)
        m_idx = np.concatenate([self.beam_indices, tk_idx], 1)
        m_fin = np.concatenate([self.is_sent_finished, just], 1)
        _, top = _topk(m_sc, K)
        self.sequences, self.beam_scores = _gather(m_seq, top), _gather(m_sc, top)
        self.beam_indices, self.is_sent_finished = _gather(m_idx, top), _gather(m_fin, top)
        beam_idx = self.running_beam_indices[..., cur - P].reshape(-1)
        self.cur_len = cur + 1
        # early-stop heuristic and loop condition
        hyp_len = (self.max_length - P) if (self.es == "never" and self.lp > 0.0) else (self.cur_len - P)
        best = self.running_beam_scores[:, :1] / f32(hyp_len ** self.lp)
        worst = np.where(self.is_sent_finished, self.beam_scores.min(axis=1, keepdims=True), NEG)
        self.unsat = self.unsat & (best > worst).any(axis=-1, keepdims=True)
        self.done = not (self.unsat.any() and not (self.is_sent_finished.all() and self.es is True) and not hits.all())
        return beam_idx

    def result(self):
        gen = int(((self.beam_indices[:, :1] + 1) != 0).sum(axis=2).max())
        return self.sequences[:, 0, :self.P + gen], self.beam_scores[:, 0]


def beam_search(score_fn, prompt, K, V, max_length, eos, length_penalty=1.0, early_stopping=False, on_step=None):
    st = BeamState(prompt, K, V, max_length, eos, length_penalty, early_stopping)
    while not st.done:
        beam_idx = st.step(score_fn(st.flat_sequences()))
        if on_step is not None:
            on_step(st, beam_idx)
    return st.result()
