"""Case tables shared by make_golden.py (generator) and the tests (consumers)."""
DLOGITS_STRIDE = 13     # big cases store every 13th gradient element
INTEGRAL_CASES = [  # name, B, J, D, H, W, logit scale
    ("tiny", 2, 3, 8, 8, 8, 3.0),
    ("rect", 3, 4, 4, 8, 16, 2.0),
    ("mid", 2, 17, 16, 16, 16, 4.0),
    ("cube32", 1, 2, 32, 32, 32, 6.0),
]
NETWORK_CASES = [  # name, layers, image, J, D, batch
    ("r18", 18, 64, 3, 8, 2),
    ("r50", 50, 128, 2, 16, 4),
]
