"""Shapes of the per-op conv fixtures (data shared by make_golden.py and the tests)."""
CONV_CASES = [
    # name, cin, cout, kernel, stride, pad, bias, act, in shape (X, Y, Z), batch
    ("k3_feature", 4, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), False, False, (6, 5, 4), 2),
    ("k3_rdb", 24, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), False, True, (5, 6, 4), 1),
    ("k1_lff_bias", 32, 16, (1, 1, 1), (1, 1, 1), (0, 0, 0), True, False, (4, 4, 5), 2),
    ("k5_hr0", 20, 20, (5, 5, 5), (1, 1, 1), (2, 2, 2), False, True, (7, 6, 5), 1),
    ("k5_hr1_bias", 20, 3, (5, 5, 5), (1, 1, 1), (2, 2, 2), True, False, (6, 7, 5), 1),
    ("k3_terrain0", 1, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), False, True, (6, 6, 4), 2),
    ("d_first", 3, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), False, True, (8, 8, 4), 2),
    ("d_down_s221", 8, 8, (4, 4, 3), (2, 2, 1), (1, 1, 1), False, True, (8, 8, 5), 2),
    ("d_down_s222", 8, 8, (4, 4, 3), (2, 2, 2), (1, 1, 1), False, True, (8, 6, 7), 2),
    ("d_s112", 16, 16, (3, 3, 3), (1, 1, 2), (1, 1, 1), False, True, (4, 4, 7), 2),
    ("k5_feat5", 8, 8, (5, 5, 5), (1, 1, 1), (2, 2, 2), False, True, (6, 6, 6), 1),
]
