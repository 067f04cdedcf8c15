"""Oracle (test infrastructure, numpy): CTC prefix scoring and the joint CTC/attention logits processor.

Restates reference src/models/dicow/decoding.py (itself derived from ESPnet's ctc_prefix_score.py, Watanabe et al. 2017
algorithm 2 in the vectorised form of Seki et al. 2019):
  * prefix_score          CTCPrefixScore.__call__ (:122-163) with _compute_log_phi (:60-72), _compute_log_psi (:74-109)
                          and _update_log_psi_with_eos (:111-120), one (sample, candidate) pair at a time, plain loops
  * initial_state         CTCPrefixScore.initial_state (:36-43)
  * CtcRescorer           CTCRescorerLogitsProcessor.__init__ / __call__ / update_state (:166-208, :263-336)

Differences in form, none in value: the reference keeps states for the whole vocabulary in a [B, V, T, 2] buffer that it
scatters into; here only the scored candidates are kept.  The reference's frame loop starts at the smallest prefix
length of the batch; starting every pair at t = 1 gives the same fp32 numbers because the forward variables of a prefix
of d labels are exactly logzero (-1e10; adding a log-probability does not change it in fp32) before frame d - 1.
Pinned against tests/golden/f13_ctc_prefix.npz (the reference classes themselves).  Only tests/ may import this module.
"""
import numpy as np

LOGZERO = np.float32(-1e10)
f32 = np.float32


def _lae(a, b):
    """torch.logaddexp in fp32."""
    a, b = f32(a), f32(b)
    m = max(a, b)
    return f32(m + np.log1p(np.exp(f32(-abs(a - b)), dtype=f32), dtype=f32))


def initial_state(x, blank):
    """x fp32 [B, T, V] log-probabilities -> r [B, T, 2] (non-blank, blank forward variables of the empty prefix)."""
    r = np.full(x.shape[:2] + (2,), LOGZERO, dtype=f32)
    r[..., 1] = np.cumsum(x[..., blank], axis=1, dtype=f32)
    return r


def prefix_score(x, rows, cs, decoded_len, last, r_prev, blank, eos):
    """x [Bx, T, V]; rows [n] sample of each hypothesis; cs [n, C] candidate labels; decoded_len [n]; last [n] last label of
    the prefix; r_prev [n, T, 2].  Returns log_psi [n, C], r [n, T, 2, C]."""
    n, C = cs.shape
    T = x.shape[1]
    psi = np.full((n, C), LOGZERO, dtype=f32)
    r = np.full((n, T, 2, C), LOGZERO, dtype=f32)
    for i in range(n):
        xi = x[rows[i]]
        r_sum = np.array([_lae(r_prev[i, t, 0], r_prev[i, t, 1]) for t in range(T)], dtype=f32)
        d = int(decoded_len[i])
        for c in range(C):
            lab = int(cs[i, c])
            xs = xi[:, lab]
            phi = r_prev[i, :, 1] if (d > 0 and lab == int(last[i])) else r_sum
            if d == 0:
                r[i, 0, 0, c] = xs[0]
            start = max(d, 1)
            acc = r[i, start - 1, 0, c]
            terms = [f32(phi[t] + xs[t + 1]) if (t + 1) >= d else LOGZERO for t in range(T - 1)]
            m = max(terms)
            lse = f32(m + np.log(np.sum(np.exp(np.array(terms, dtype=f32) - m, dtype=f32), dtype=f32), dtype=f32))
            acc = _lae(acc, lse)
            for t in range(1, T):
                r[i, t, 0, c] = f32(_lae(r[i, t - 1, 0, c], phi[t - 1]) + xs[t])
                r[i, t, 1, c] = f32(_lae(r[i, t - 1, 0, c], r[i, t - 1, 1, c]) + xi[t, blank])
            if lab == eos:
                acc = r_sum[T - 1]
            elif lab == blank and eos != blank:
                acc = LOGZERO
            psi[i, c] = acc
    return psi, r


class CtcRescorer:
    """decoding.py:166-208 (__init__), :263-336 (__call__), :254-261 (update_state); greedy or beam rows alike."""

    def __init__(self, encoder_logits, blank, eos, bos, first_timestamp, upper_cased, prefix_len, ctc_weight, n_score=500):
        z = encoder_logits.astype(f32)
        m = z.max(-1, keepdims=True)
        lp = (z - m) - np.log(np.exp(z - m, dtype=f32).sum(-1, keepdims=True, dtype=f32), dtype=f32)      # log_softmax
        for lo, up in upper_cased:
            lp[..., up] = lp[..., lo]
        self.x, self.blank, self.eos, self.bos, self.ts0 = lp.astype(f32), blank, eos, bos, first_timestamp
        self.prefix_len, self.w, self.k = prefix_len, f32(ctc_weight), n_score
        self.V = lp.shape[-1] - 1
        self.state_prev = initial_state(self.x, blank)
        self.score_prev = np.zeros((lp.shape[0], 1), dtype=f32)
        self.cand = self.cand_scores = self.cand_states = None

    def __call__(self, input_ids, scores):
        ids = np.array(input_ids).copy()
        if (ids[:, 0] != self.bos).any():
            ids = np.stack([row[int(np.nonzero(row == self.bos)[0][0]):] for row in ids])
        if self.prefix_len > 1:
            ids = ids[:, self.prefix_len - 1:]
        ids[:, 0] = self.blank
        decoded_len = ((ids <= self.ts0) & (ids != self.blank)).sum(1)
        is_ts = (ids[:, -1] >= self.ts0) & (ids[:, -1] != self.blank)
        pos = ((ids < self.ts0) | (ids == self.blank)).sum(1) - 1
        repl = ids[np.arange(len(ids)), pos]
        ids[is_ts, -1] = repl[is_ts]
        todo = ids[:, -1] != self.eos
        B = len(ids)
        head = scores[:, :self.ts0]
        cand = np.argsort(-head, axis=1, kind="stable")[:, :self.k]
        for b in range(B):
            if self.eos not in cand[b]:
                cand[b, self.k - 1] = self.eos
        full = np.full((B, self.V), LOGZERO, dtype=f32)
        states = np.full((B,) + self.state_prev.shape[1:] + (self.k,), LOGZERO, dtype=f32)
        rows = np.nonzero(todo)[0]
        if len(rows):
            psi, r = prefix_score(self.x, rows, cand[rows], decoded_len[rows], ids[rows, -1], self.state_prev[rows], self.blank, self.eos)
            for j, b in enumerate(rows):
                full[b, cand[b]] = psi[j]
                states[b] = r[j]
        self.cand, self.cand_scores, self.cand_states = cand, full.copy(), states
        full[:, self.ts0:] = full.max(axis=1)[:, None]
        self.full = full
        return ((f32(1) - self.w) * scores.astype(f32) + self.w * (full - self.score_prev)).astype(f32)

    def update_state(self, best_ids, beam_idx):
        new_state, new_score = self.state_prev[beam_idx].copy(), self.score_prev[beam_idx].copy()
        for j, (b, tok) in enumerate(zip(beam_idx, best_ids)):
            if tok < self.ts0:
                hit = np.nonzero(self.cand[b] == tok)[0]
                if len(hit):
                    new_state[j] = self.cand_states[b][..., hit[0]]
                new_score[j, 0] = self.full[b, tok]
        self.state_prev, self.score_prev = new_state, new_score
