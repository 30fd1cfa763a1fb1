"""Token vocabularies of the two tracks on the hot path.

[ESM-RECALL] values of esm.utils.constants.esm3 (esm==3.0.4, not vendored in the reference), cross-checked
against the reference's own use: vocab 4096 + 5 specials (/root/reference/slm/models/model.py:380-381),
pad id 4099 (configs/model/default.yaml:39-41), '_' as the mask residue (slm/models/utils.py:121).
"""
VQVAE_CODEBOOK_SIZE = 4096
STRUCTURE_MASK_TOKEN = 4096
STRUCTURE_EOS_TOKEN = 4097
STRUCTURE_BOS_TOKEN = 4098
STRUCTURE_PAD_TOKEN = 4099
STRUCTURE_CHAINBREAK_TOKEN = 4100
STRUCTURE_VOCAB = 4101

SEQUENCE_BOS_TOKEN = 0
SEQUENCE_PAD_TOKEN = 1
SEQUENCE_EOS_TOKEN = 2
SEQUENCE_CHAINBREAK_TOKEN = 31
SEQUENCE_MASK_TOKEN = 32
SEQUENCE_VOCAB = [
    "<cls>", "<pad>", "<eos>", "<unk>",
    "L", "A", "G", "V", "S", "E", "R", "T", "I", "D", "P", "K", "Q", "N", "F", "Y", "M", "H", "W", "C",
    "X", "B", "U", "Z", "O", ".", "-", "|", "<mask>",
]
MASK_RESIDUE = "_"
