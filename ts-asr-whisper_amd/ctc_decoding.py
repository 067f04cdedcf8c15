"""Joint CTC / attention decoding on the GPU: CTC prefix scoring + the logits processor.

Mirrors reference src/models/dicow/decoding.py:
  * ``CtcPrefixScorer``   CTCPrefixScore (:8-163): initial state and the per-token scoring of candidate labels -- the frame
                          recursion runs in ONE kernel (csrc/ctc_prefix.hip) instead of a Python loop over the frames;
  * ``CtcRescorer``       CTCRescorerLogitsProcessor (:166-208 construction, :263-336 __call__, :254-261 update_state);
  * ``log_softmax_scores``  LogSoftmaxProcessor (:339-349), applied before the rescorer in greedy search (generation.py:252).

The glue (top-k candidate selection, scatter of the candidate scores into the vocabulary row, state selection) is a handful
of small torch tensor ops on [B, V]; states are kept for the scored candidates only ([B, T, 2, k]) where the reference
scatters them into a [B, V, T, 2] buffer.  Everything stays on the device with fixed shapes (finished rows are scored
and then masked instead of being filtered out); after the first call, which inspects the prompt layout, a decoding step adds
no host synchronisation.  No CPU fallback.
"""
import torch

from . import _lib as L

F32, BF16, I32 = torch.float32, torch.bfloat16, torch.int32
LOGZERO = -1e10


def log_softmax_scores(scores):
    return torch.log_softmax(scores.float(), dim=-1)


class CtcPrefixScorer:
    """enc_logits [B, T, V1] fp32 / bf16 on the GPU (rows may be padded: any stride(1) >= V1), blank = V1 - 1 by default."""

    def __init__(self, enc_logits, blank, eos, alias=None):
        if not enc_logits.is_cuda:
            raise L.DicowError("CtcPrefixScorer: tensors must be on the GPU (no CPU fallback)")
        if enc_logits.dtype not in (F32, BF16) or enc_logits.stride(2) != 1 or enc_logits.stride(0) != enc_logits.shape[1] * enc_logits.stride(1):
            raise L.DicowError("CtcPrefixScorer: expected fp32/bf16 frame-major logits [B, T, V1] with dense frames")
        self.x, self.blank, self.eos = enc_logits, int(blank), int(eos)
        self.B, self.T, self.V1 = enc_logits.shape
        self.ld, self.bf16 = enc_logits.stride(1), int(enc_logits.dtype == BF16)
        dev = enc_logits.device
        self.alias = None if alias is None else alias.to(device=dev, dtype=I32).contiguous()
        self.lse = torch.empty(self.B * self.T, dtype=F32, device=dev)
        L.call("dicow_ctc_frame_lse", self.x.data_ptr(), self.bf16, self.B * self.T, self.V1, self.ld, self.lse.data_ptr(), L.stream())

    def initial_state(self):
        r0 = torch.empty(self.B, self.T, 2, dtype=F32, device=self.x.device)
        col = self.blank if self.alias is None else int(self.alias[self.blank])
        L.call("dicow_ctc_prefix_init", self.x.data_ptr(), self.bf16, self.ld, self.lse.data_ptr(), self.B, self.T, col, r0.data_ptr(),
               L.stream())
        return r0

    def __call__(self, rows, cs, decoded_len, last, r_prev):
        """rows [n], cs [n, C], decoded_len [n], last [n] (any integer dtype, on the device), r_prev fp32 [n, T, 2]
        -> psi fp32 [n, C], r fp32 [n, T, 2, C]."""
        n, C = cs.shape
        dev = self.x.device
        a = L.CtcPrefixArgs()
        keep = [t.to(device=dev, dtype=I32).contiguous() for t in (rows, cs, decoded_len, last)]
        r_prev = r_prev.to(F32).contiguous()
        psi = torch.empty(n, C, dtype=F32, device=dev)
        r = torch.empty(n, self.T, 2, C, dtype=F32, device=dev)
        a.logits, a.in_bf16, a.ld, a.lse = self.x.data_ptr(), self.bf16, self.ld, self.lse.data_ptr()
        a.alias = None if self.alias is None else self.alias.data_ptr()
        a.rows, a.cs, a.decoded_len, a.last = (t.data_ptr() for t in keep)
        a.r_prev, a.psi, a.r = r_prev.data_ptr(), psi.data_ptr(), r.data_ptr()
        a.n, a.C, a.T, a.blank, a.eos = n, C, self.T, self.blank, self.eos
        L.call_struct("dicow_ctc_prefix_score", a)
        return psi, r


class CtcRescorer:
    """next_scores = (1 - w) * scores + w * (ctc prefix score of prefix+token - ctc score of the prefix).

    enc_logits [B, T, V + 1] (blank last); ``upper_cased``: iterable of (lower_id, upper_id) pairs whose CTC columns are tied
    (tokenizer.upper_cased_tokens.items()); ``prefix_len`` = len(tokenizer.prefix_tokens); ``first_timestamp`` =
    tokenizer.get_vocab()["<|0.00|>"]; rows of a beam search are hypotheses (repeat the logits num_beams times)."""

    def __init__(self, enc_logits, blank, eos, bos, first_timestamp, upper_cased, prefix_len, ctc_weight, n_score=500, num_beams=1):
        dev = enc_logits.device
        V1 = enc_logits.shape[-1]
        alias = torch.arange(V1, dtype=I32)
        for lo, up in upper_cased:
            alias[int(up)] = int(lo)
        self.scorer = CtcPrefixScorer(enc_logits, blank, eos, alias.to(dev))
        # hypotheses = batch rows x beams (beam-major within a row, like HF's flattened beam dimension); the reference repeats
        # the logits num_beams times (generation.py:257), here every hypothesis just points at its batch row
        self.B, self.T, self.V = enc_logits.shape[0] * int(num_beams), enc_logits.shape[1], V1 - 1
        self.blank, self.eos, self.bos, self.ts0 = int(blank), int(eos), int(bos), int(first_timestamp)
        self.prefix_len, self.w, self.k = int(prefix_len), float(ctc_weight), int(n_score)
        self.rows = torch.arange(self.B, device=dev) // int(num_beams)
        self.state_prev = self.scorer.initial_state()[self.rows]
        self.score_prev = torch.zeros(self.B, 1, dtype=F32, device=dev)
        self.cand = self.cand_states = self.full = None
        self._cut = None

    def __call__(self, input_ids, scores):
        ids = input_ids.clone()
        if self._cut is None:                         # decoding.py:266-269: drop whatever precedes bos (checked once: the
            first = ids[:, 0].tolist()                # prompt layout does not change while a window is being decoded)
            if all(t == self.bos for t in first):
                self._cut = 0
            else:
                cut = [int((row == self.bos).nonzero()[0]) for row in ids]
                if len(set(cut)) != 1:
                    raise L.DicowError("CtcRescorer: rows carry prompts of different lengths before bos")
                self._cut = cut[0]
        if self._cut:
            ids = ids[:, self._cut:].clone()
        if self.prefix_len > 1:
            ids = ids[:, self.prefix_len - 1:].clone()
        ids[:, 0] = self.blank
        not_blank = ids != self.blank
        decoded_len = ((ids <= self.ts0) & not_blank).sum(1)
        last = ids[:, -1]
        is_ts = (last >= self.ts0) & (last != self.blank)
        pos = ((ids < self.ts0) | ~not_blank).sum(1, keepdim=True) - 1
        last = torch.where(is_ts, ids.gather(1, pos)[:, 0], last)
        todo = last != self.eos
        scores = scores.float()
        cand = torch.topk(scores[:, :self.ts0], k=self.k).indices
        has_eos = (cand == self.eos).any(dim=1)
        cand[:, self.k - 1] = torch.where(has_eos, cand[:, self.k - 1], torch.full_like(cand[:, 0], self.eos))
        psi, r = self.scorer(self.rows, cand, decoded_len, last, self.state_prev)
        full = torch.full((self.B, self.V), LOGZERO, dtype=F32, device=scores.device)
        full.scatter_(1, cand, psi)
        full = torch.where(todo[:, None], full, torch.full_like(full, LOGZERO))     # rows that ended with eos are not scored
        full[:, self.ts0:] = full.max(dim=1).values[:, None]                        # timestamps: neutral (the row maximum)
        self.cand, self.cand_states, self.full = cand, r, full
        return (1.0 - self.w) * scores + self.w * (full - self.score_prev)

    def update_state(self, best_ids, beam_idx):
        cand, states = self.cand[beam_idx], self.cand_states[beam_idx]
        hit = cand == best_ids[:, None]
        idx = hit.float().argmax(dim=1)
        picked = states.gather(3, idx.view(-1, 1, 1, 1).expand(-1, self.T, 2, 1))[..., 0]
        take = (best_ids < self.ts0) & hit.any(dim=1)
        self.state_prev = torch.where(take[:, None, None], picked, self.state_prev[beam_idx])
        score = self.full[beam_idx].gather(1, best_ids.clamp(max=self.V - 1)[:, None])
        self.score_prev = torch.where((best_ids < self.ts0)[:, None], score, self.score_prev[beam_idx])
