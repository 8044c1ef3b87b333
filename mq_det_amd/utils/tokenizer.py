"""Synthetic `bert-base-uncased`-shaped tokenizer directory (no network / vocab files in the image).

The reference calls `AutoTokenizer.from_pretrained(cfg.MODEL.LANGUAGE_BACKBONE.TOKENIZER_TYPE)`
(generalized_vl_rcnn_new.py:144, engine/inference.py:259-263) and accepts a local path whose basename is
"bert-base-uncased".  This writes such a directory with a 30522-entry WordPiece vocab: BERT's special-token ids
([PAD]=0, [UNK]=100, [CLS]=101, [SEP]=102, [MASK]=103), punctuation, letters / '##' pieces and synthetic words.
"""
import json
import os
import string


def synthetic_vocab(size=30522, extra_words=()):
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    vocab += [f"[unused{i}]" for i in range(99, 99 + 1000 - len(vocab))]
    vocab += list(string.punctuation) + list(string.digits) + list(string.ascii_lowercase)
    vocab += ["##" + c for c in string.digits + string.ascii_lowercase]
    vocab += [w for w in dict.fromkeys(extra_words) if w not in vocab]           # whole words (e.g. real category names)
    i = 0
    while len(vocab) < size:
        vocab.append(f"obj{i}")
        i += 1
    return vocab[:size]


def build_synthetic_tokenizer(root, size=30522, extra_words=()):
    """Creates <root>/bert-base-uncased/{vocab.txt,tokenizer_config.json} and returns that path."""
    path = os.path.join(root, "bert-base-uncased")
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "vocab.txt"), "w") as f:
        f.write("\n".join(synthetic_vocab(size, extra_words)) + "\n")
    with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
        json.dump({"do_lower_case": True, "tokenizer_class": "BertTokenizer", "model_max_length": 512}, f)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump({"model_type": "bert", "vocab_size": size}, f)
    return path


def synthetic_caption(num_classes, start=0, sep=". ", words=(1, 2, 3, 4, 3, 2)):
    """Synthetic class-name caption + char spans, the shape create_queries_and_maps builds (engine/inference.py:212-283).
    Class i has words[i % len(words)] words ('obj7 obj8 obj9'), like LVIS names that split into several word pieces: the
    default gives 2.5 tokens + 1 separator per class = 142 tokens for 40 classes (120-200 tokens
    for an LVIS chunk); words=(1,) is the 81-token 'obj0. obj1. ...' caption of round 1."""
    caption, spans, w = "", [], start
    for i in range(num_classes):
        n = words[i % len(words)]
        name = " ".join(f"obj{w + j}" for j in range(n))
        w += n
        spans.append((len(caption), len(caption) + len(name)))
        caption += name
        if i != num_classes - 1:
            caption += sep
    return caption, spans


def positive_map_from_spans(tokenizer, caption, spans, labels):
    """engine/inference.py:130-163 (create_positive_dict): {label: [token idx]} via char_to_token."""
    tok = tokenizer(caption, return_tensors="pt")
    pmap = {}
    for (beg, end), lab in zip(spans, labels):
        b, e = tok.char_to_token(beg), tok.char_to_token(end - 1)
        if b is None or e is None:
            continue
        pmap[lab] = list(range(b, e + 1))
    return pmap
