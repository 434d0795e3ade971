"""CPU: the index arithmetic of the direct-to-LDS operand layouts of csrc/gemm_pipe.hip, modelled in Python.

A `buffer_load_dwordx4 ... lds` writes lane l's 16 bytes to LDS at base + 16 l, so the kernel cannot pad rows; it stores the
tiles unpadded with an XOR swizzle and lets every lane FETCH the chunk that belongs at its slot.  These tests restate the
formulas of the kernel (loader side and fragment-read side) and check that (1) the loader fills every slot with a distinct
chunk of the right row, (2) the fragment reads find exactly the value the MFMA step needs, (3) the reads of one wave access
are spread over the LDS banks as well as the padded layout spread them.  (The device check is tools/gemm_ldsdma_check.py:
bitwise equality with the register-staged kernel.)"""
import itertools

BM, BK = 128, 32


def test_plain_tile_loader_and_fragment_reads_agree():
    # loader: thread tid -> srow = tid >> 3, kq = tid & 7; instruction i covers tile rows srow + 32 i; the wave's LDS base
    # is row 8 * wave + 32 i, lane l lands at + 16 l bytes = row (l >> 3), slot (l & 7); it fetches chunk kq ^ (srow & 7)
    lds = {}                                              # (row, slot) -> (row, k-chunk) stored there
    for tid, i in itertools.product(range(256), range(4)):
        wave, lane = tid >> 6, tid & 63
        srow, kq = tid >> 3, tid & 7
        row = 8 * wave + 32 * i + (lane >> 3)
        slot = lane & 7
        assert row == srow + 32 * i and slot == kq
        chunk = kq ^ (srow & 7)
        assert (row, slot) not in lds
        lds[(row, slot)] = (row, chunk)
    assert len(lds) == BM * 8
    for row in range(BM):
        assert sorted(c for (r, s), (_, c) in lds.items() if r == row) == list(range(8))
    # fragment read of MFMA quarter q: lane (r, half) of a wave whose block starts at row rb (a multiple of 32) wants k-chunk
    # 2 q + half of rows rb + r and rb + r + 32: slot (2 q + half) ^ (r & 7)
    for rb, r, half, q in itertools.product((0, 32, 64, 96), range(32), range(2), range(4)):
        for extra in (0, 32):
            row = rb + r + extra
            if row >= BM:
                continue
            slot = (2 * q + half) ^ (r & 7)
            assert lds[(row, slot)] == (row, 2 * q + half)


def test_plain_tile_fragment_reads_are_bank_conflict_free():
    # a ds_read_b128 is served 8 lanes (128 bytes) at a time; the 8 lanes r .. r + 7 of one half must hit 8 different 16-byte
    # columns of the 128-byte-wide bank array (unpadded rows are 128 bytes: the column IS the slot)
    for half, q, r0 in itertools.product(range(2), range(4), range(0, 32, 8)):
        cols = {((2 * q + half) ^ (r & 7)) for r in range(r0, r0 + 8)}
        assert len(cols) == 8


def test_kstrided_tile_loader_and_fragment_reads_agree():
    # K-strided tile: 32 memory rows (k) of 128 floats (32 chunks).  Loader: tk = tid >> 5, tc = tid & 31, instruction i covers
    # k row tk + 8 i; wave base = k row 2 * wave + 8 i, lane l at + 16 l bytes = k row (l >> 5), slot (l & 31); it fetches
    # column chunk tc ^ 8 * ((tk >> 2) & 1)
    lds = {}
    for tid, i in itertools.product(range(256), range(4)):
        wave, lane = tid >> 6, tid & 63
        tk, tc = tid >> 5, tid & 31
        krow = 2 * wave + 8 * i + (lane >> 5)
        slot = lane & 31
        assert krow == tk + 8 * i and slot == tc
        chunk = tc ^ (((tk >> 2) & 1) * 8)
        assert ((krow >> 2) & 1) == ((tk >> 2) & 1)       # the swizzle bit of the k row does not depend on i
        assert (krow, slot) not in lds
        lds[(krow, slot)] = (krow, chunk)
    assert len(lds) == BK * 32
    # fragment read (ds_read_b32) of step t of quarter q: lane (r, half) wants the value of k row 8 q + 4 half + t at column
    # m = rb + r (and m + 32): float address j * 128 + 4 * ((m >> 2) ^ 8 half) + (m & 3)
    for rb, r, half, q, t in itertools.product((0, 32, 64), range(32), range(2), range(4), range(4)):
        j = 8 * q + 4 * half + t
        assert ((j >> 2) & 1) == half
        for extra in (0, 32):
            m = rb + r + extra
            slot = (m >> 2) ^ (half * 8)
            assert lds[(j, slot)] == (j, m >> 2)
    # one wave access (64 lanes x 4 bytes): the two halves read k rows 4 apart; with unpadded 512-byte rows they would hit
    # the same 32 banks - the swizzle moves half 1 by 8 chunks = 32 banks
    for rb, q, t in itertools.product((0, 64), range(4), range(4)):
        banks = [(((rb + r) >> 2) ^ (half * 8)) * 4 + ((rb + r) & 3) for half in range(2) for r in range(32)]
        assert len({b % 64 for b in banks}) == 64


def test_shifted_tail_tile_covers_every_k_once():
    """K % 32 == 16 on the direct-to-LDS path (experimental): full tiles cover k < K - 16; the last tile is fetched from
    K - 32 and only its quarters 2 and 3 (columns 16 .. 31 of the tile) are multiplied: every k exactly once, ascending."""
    for K in (48, 176, 208, 1040):
        nfull = K // 32
        order = []
        for kt in range(nfull):
            for q in range(4):
                order += [kt * 32 + 8 * q + j for j in range(8)]
        base = K - 32                                  # the tail tile's first column (inside the operand row)
        for q in (2, 3):
            order += [base + 8 * q + j for j in range(8)]
        assert order == list(range(K))
