"""Host-side arithmetic that mirrors device-side index logic (no GPU): properties the kernels rely on."""
def test_xcd_tile_order_is_a_bijection():
    """csrc/gemm.hip xcd_tile: workgroup L (dealt to XCD L % 8) takes tile start_k + L // 8 of the contiguous range of XCD k = L % 8.
    The same arithmetic in Python: every tile is taken exactly once for any tile count, and the tiles of one XCD are consecutive."""
    for T in list(range(16, 200)) + [223, 256, 1000, 6400, 12345]:
        q, r = T >> 3, T & 7
        seen = [0] * T
        per = {}
        for L in range(T):
            k, j = L & 7, L >> 3
            t = k * q + min(k, r) + j
            assert 0 <= t < T
            seen[t] += 1
            per.setdefault(k, []).append(t)
        assert all(c == 1 for c in seen), T
        for k, ts in per.items():
            assert ts == list(range(ts[0], ts[0] + len(ts)))
