"""Decode post-processing and the BLEU variant the reference reports (train.py:112-119,
evaluate.py:101-116).

``sentence_bleu`` restates nltk.translate.bleu_score.sentence_bleu (nltk 3.x defaults: no
smoothing function = method0, ``auto_reweigh=False``).  nltk is not installed in the build
image; the restatement follows the published algorithm (modified n-gram precision,
closest-reference-length brevity penalty, zero precisions replaced by ``sys.float_info.min`` with
a warning in nltk) and is pinned to the exact values nltk's own doctests publish
(``sentence_bleu`` 0.5045666840058485 and 0.3920, ``corpus_bleu``'s 0.6223..., the
``modified_precision`` fractions — tests/test_host_cpu.py).  The reference calls it with the list
of reference WORDS as the list of references (train.py:115 passes ``question_str_list``), so each
"reference" is a single word that is iterated character by character — ``reference_style=True``
reproduces exactly that call; ``False`` computes standard single-reference sentence BLEU."""
from __future__ import annotations

import math
import sys
from collections import Counter
from typing import List, Sequence


def truncate_at_end(ids: Sequence[int], end_id: int) -> List[int]:
    """evaluate.py:101-103: stop at the first <end> and drop it."""
    out = []
    for t in ids:
        if int(t) == end_id:
            break
        out.append(int(t))
    return out


def ids_to_words(ids: Sequence[int], index_to_word: dict) -> List[str]:
    """The reference keys index_to_word.json by str(index) (train.py:109)."""
    return [index_to_word[str(int(i))] for i in ids]


def _ngrams(seq, n):
    return [tuple(seq[i:i + n]) for i in range(len(seq) - n + 1)]


def sentence_bleu(references: Sequence[Sequence], hypothesis: Sequence, weights=(0.25, 0.25, 0.25, 0.25)) -> float:
    hyp = list(hypothesis)
    refs = [list(r) for r in references]
    p_num, p_den = [], []
    for n in range(1, len(weights) + 1):
        counts = Counter(_ngrams(hyp, n))
        max_counts = {}
        for r in refs:
            rc = Counter(_ngrams(r, n))
            for g in counts:
                max_counts[g] = max(max_counts.get(g, 0), rc[g])
        p_num.append(sum(min(c, max_counts[g]) for g, c in counts.items()))
        p_den.append(max(1, sum(counts.values())))
    if p_num[0] == 0:
        return 0.0
    hyp_len = len(hyp)
    ref_len = min((len(r) for r in refs), key=lambda rl: (abs(rl - hyp_len), rl))
    bp = 1.0 if hyp_len > ref_len else (0.0 if hyp_len == 0 else math.exp(1 - ref_len / hyp_len))
    s = 0.0
    for w, num, den in zip(weights, p_num, p_den):
        s += w * math.log(num / den if num > 0 else sys.float_info.min)
    return bp * math.exp(s)


def reference_bleu_scores(question: str, pred_words: Sequence[str]) -> dict:
    """The four numbers validate()/evaluate() accumulate (train.py:115-119): the reference
    question's word list is passed as the list of references."""
    refs = question.split()
    return {"bleu_1": sentence_bleu(refs, pred_words, (1, 0, 0, 0)),
            "bleu_2": sentence_bleu(refs, pred_words, (0.5, 0.5, 0, 0)),
            "bleu_3": sentence_bleu(refs, pred_words, (0.33, 0.33, 0.33, 0)),
            "bleu": sentence_bleu(refs, pred_words)}
