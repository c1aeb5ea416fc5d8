"""Special token ids (mirror of /root/reference/neuroir/inputters/constants.py:1-9)."""
PAD, UNK, BOS, EOS = 0, 1, 2, 3
PAD_WORD, UNK_WORD, BOS_WORD, EOS_WORD = "<blank>", "<unk>", "<s>", "</s>"
