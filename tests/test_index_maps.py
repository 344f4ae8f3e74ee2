"""Index arithmetic of the persistent conv kernels, restated in numpy (no GPU): the XCD-aware band walk and the LDS bank maps of
the channel-innermost patches (csrc/conv_tc.h `first_tile`, `DCS`, `CS`; csrc/conv3_tc.h `F_CS`).  The kernels carry the same
facts as comments / static_asserts; these tests pin the numbers the comments quote."""
import numpy as np
import pytest


def first_tile(g, grid, bpi):
    """csrc/conv_tc.h first_tile<BPI>: first band of workgroup g; a workgroup then walks tile += grid."""
    if grid != 256:
        return g
    x, h = g & 7, g >> 3
    return (8 * (h // bpi) + x) * bpi + h % bpi


@pytest.mark.parametrize("bpi", [8, 16])
def test_band_walk_is_a_bijection_and_keeps_an_image_on_one_xcd(bpi):
    starts = np.array([first_tile(g, 256, bpi) for g in range(256)])
    assert sorted(starts.tolist()) == list(range(256))                 # one sweep covers 256 consecutive tiles exactly once
    for n_img in (32, 40, 480):
        seen = {}
        for g in range(256):
            t = first_tile(g, 256, bpi)
            while t < n_img * bpi:
                assert t not in seen
                seen[t] = g
                t += 256
        assert len(seen) == n_img * bpi                                # every band of every image is some workgroup's
        for img in range(n_img):
            xcds = {seen[img * bpi + b] & 7 for b in range(bpi)}       # workgroup g runs on XCD g % 8 (round-robin dispatch)
            assert len(xcds) == 1
    for grid in (8, 24, 255):                                          # small batches: the plain walk
        assert [first_tile(g, grid, bpi) for g in range(grid)] == list(range(grid))


def _groups_b128():
    """lane groups of a ds_read_b128 (MI355X_MICROARCH.md, LDS table)"""
    base = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    return [[lane + 32 * half for lane in g] for half in (0, 1) for g in base]


def test_dgrad_dy_patch_reads_are_conflict_free():
    # conv12 data gradient, [row][col][co 48 + 2]: lane (lr = column, lq = co % 4) reads word (col * 50 + lq); a ds_read2_b32 is
    # two b32 reads, each serviced in two 32-lane halves over 32 banks
    dcs = 50
    for half in (0, 1):
        banks = [((lane & 15) * dcs + (lane >> 4)) % 32 for lane in range(32 * half, 32 * half + 32)]
        assert len(set(banks)) == 32
    # the pooled cells' un-pooled stores: a 32-lane half = 8 px x 4 co, a pooled px = 2 positions = 100 words
    banks = [((2 * (lane & 7)) * dcs + ((lane >> 3) & 3)) % 32 for lane in range(32)]
    assert len(set(banks)) == 32


def test_forward_patch_b64_reads_cover_the_64_banks_once():
    # conv12 forward, [row][col][ci 32 + 2]: lane (lr, lq) reads the two words at (2 lr) * 34 + 2 lq (stride-2 convolution)
    cs = 34
    for half in (0, 1):
        words = []
        for lane in range(32 * half, 32 * half + 32):
            w = 2 * (lane & 15) * cs + 2 * (lane >> 4)
            words += [w % 64, (w + 1) % 64]
        assert len(set(words)) == 64
    assert 2 * 9 * 65 * cs * 4 <= 160 * 1024                           # two patches fit the CU's LDS


def test_conv3_forward_patch_b128_reads_are_conflict_free():
    # conv3 forward, [row 9][col 24][ci 48 + 4]: lane (lr = (oy & 1) * 8 + ox, lq) reads 4 words at the position's base + 4 lq
    rs, cs = 24, 52
    for grp in _groups_b128():
        words = []
        for lane in grp:
            lr, lq = lane & 15, lane >> 4
            w = ((2 * (lr >> 3)) * rs + 2 * (lr & 7)) * cs + 4 * lq
            words += [(w + q) % 64 for q in range(4)]
        assert len(set(words)) == 64
    # staging thread = (column quad tid % 4, channel 8 (tid / 32 % 8) + tid / 4 % 8, row phase tid / 256), items = rows 5 phase + j:
    # a 32-lane half holds 4 quads x 8 channels of one row - its transposing stores are 2-way at worst
    for half in range(16):
        banks = {}
        for tid in range(32 * half, 32 * half + 32):
            c4, ci, r = tid & 3, 8 * ((tid >> 5) & 7) + ((tid >> 2) & 7), 5 * (tid >> 8)
            banks.setdefault(((r * rs + 1 + 4 * c4) * cs + ci) % 32, []).append(tid)
        assert max(len(v) for v in banks.values()) <= 2
    # ... and the items cover the unit's 48 x 9 x 4 float4 exactly once
    items = []
    for tid in range(512):
        c4, ci, r0 = tid & 3, 8 * ((tid >> 5) & 7) + ((tid >> 2) & 7), 5 * (tid >> 8)
        items += [(ci, r0 + j, c4) for j in range(5) if ci < 48 and r0 + j < 9]
    assert len(items) == 1728 and len(set(items)) == 1728


def test_forward_k_step_order_matches_the_weight_registers():
    # conv12 forward: k-step ks = (tap, pair m, j) takes ci = 8 m + 2 lq + j; over the 8 k-steps of a tap and the 4 lane groups
    # every input channel appears exactly once.  conv3 forward: ks = 12 tap + 4 g + j takes ci = 16 g + 4 lq + j (48 channels).
    for tap in range(9):
        ci = sorted(8 * ((ks & 7) >> 1) + 2 * lq + (ks & 1) for ks in range(8 * tap, 8 * tap + 8) for lq in range(4))
        assert ci == list(range(32))
        ci3 = sorted(16 * ((ks % 12) >> 2) + 4 * lq + (ks & 3) for ks in range(12 * tap, 12 * tap + 12) for lq in range(4))
        assert ci3 == list(range(48))


def test_conv3_forward_weight_staging_is_tap_major_and_conflict_free():
    # csrc/conv3_tc.h conv3w_stage_tap_major<512, 472, 52>: LDS word of W[co][ci][tap] = co * 472 + tap * 52 + ci; a lane (lr -> co,
    # lq) gathers the four channels 16 g + 4 lq + j of k-steps (tap, g, j) with ONE ds_read_b128 (27 per lane for its 108 weights)
    S, TS = 472, 52
    for grp in _groups_b128():
        words = []
        for lane in grp:
            lr, lq = lane & 15, lane >> 4
            words += [(lr * S + 4 * lq + q) % 64 for q in range(4)]
        assert len(set(words)) == 64
    assert 64 * S * 4 <= 160 * 1024 and 9 * TS <= S and TS % 4 == 0 and S % 4 == 0       # fits, rows do not overlap, 16-byte reads
    # staging items (co, ci) = tid + 512 j: every weight lands exactly once ...
    seen = set()
    for i in range(64 * 48):
        co, ci = divmod(i, 48)
        for t in range(9):
            seen.add(co * S + t * TS + ci)
    assert len(seen) == 64 * 432
    # ... and the 32 lanes of a store (consecutive items, one tap) are at most 2-way on the 32 banks (the row seam: 472 = 24 mod 32)
    worst = 0
    for base in range(0, 64 * 48, 32):
        banks = {}
        for i in range(base, base + 32):
            co, ci = divmod(i, 48)
            banks[(co * S + ci) % 32] = banks.get((co * S + ci) % 32, 0) + 1
        worst = max(worst, max(banks.values()))
    assert worst <= 2
    # the gather's k-step order: quad qd = 3 tap + g -> registers 4 qd + j = 12 tap + 4 g + j, the order the MFMA loop consumes
    assert [(4 * qd + j) for qd in range(27) for j in range(4)] == [12 * (qd // 3) + 4 * (qd % 3) + j for qd in range(27) for j in range(4)]


# ---- the split-precision conv12 kernels (csrc/conv_split.h) and the encoder Linear forward (csrc/enc_linear.h) ----------------------
def _b128_worst(addr):
    """largest number of distinct 16-byte reads that share a bank inside one 16-lane group of a ds_read_b128 (1 = conflict-free);
    addr(lane) = first dword of the lane's read, banks = dword mod 64"""
    worst = 0
    for g in _groups_b128():
        per_bank = {}
        for lane in g:
            a = addr(lane)
            for k in range(4):
                per_bank.setdefault((a + k) % 64, set()).add(a)
        worst = max(worst, max(len(v) for v in per_bank.values()))
    return worst


def _b32_store_worst(addr):
    """the same for a ds_write_b32: two 32-lane halves, banks = dword mod 32"""
    worst = 0
    for half in (range(32), range(32, 64)):
        per_bank = {}
        for lane in half:
            per_bank.setdefault(addr(lane) % 32, set()).add(addr(lane))
        worst = max(worst, max(len(v) for v in per_bank.values()))
    return worst


def test_split_dgrad_dy_fragments_and_cell_stores():
    # dY piece plane [row][col 33][co 48] bf16, dense: 24 dwords per position; lane (lr = position, lq = k-group of 8 channels = 4 dwords)
    assert _b128_worst(lambda lane: 24 * (lane & 15) + 4 * (lane >> 4)) == 1
    # a padded position (26 / 28 dwords) would not be: 24 = 8 mod 16 is what the lane groups of the read want
    assert _b128_worst(lambda lane: 28 * (lane & 15) + 4 * (lane >> 4)) > 1
    # cell pairs: a 32-lane half = 8 channel pairs x 4 px; dword (2 px) * 24 + pair -> 2 lanes per bank (free for a ds_write_b32) ...
    assert _b32_store_worst(lambda lane: 48 * ((lane >> 3) & 7) + (lane & 7)) == 2
    # ... where px fastest over 16 lanes put 8 on a bank
    assert _b32_store_worst(lambda lane: 48 * (lane & 15) + (lane >> 4)) == 8


def test_split_wgrad_patch_and_dy_layouts():
    sw = lambda ch: (ch >> 2) & 3                                                                  # noqa: E731  granule XOR of a channel
    # A fragments: lane (lr = channel of the half, lq = position group) reads granule lq ^ sw(lr) of its channel's 64-byte row
    assert _b128_worst(lambda lane: 488 * (lane & 15) + 4 * ((lane >> 4) ^ sw(lane & 15))) == 1
    assert _b128_worst(lambda lane: 488 * (lane & 15) + 4 * (lane >> 4)) == 1                      # (the XOR is for the stores)
    assert _b128_worst(lambda lane: 484 * (lane & 15) + 4 * (lane >> 4)) > 1                       # 36 mod 64: the first layout, 41 % conflicts
    # conv1's stores: lane (lr = channel, lq = column quad) writes dword lq of granule cg ^ sw(lr)
    for cg in range(4):
        assert _b32_store_worst(lambda lane, cg=cg: 488 * (lane & 15) + 4 * (cg ^ sw(lane & 15)) + (lane >> 4)) == 2
        assert _b32_store_worst(lambda lane, cg=cg: 488 * (lane & 15) + 4 * cg + (lane >> 4)) == 4  # without the XOR
    # dY: 40 dwords per output channel; B fragments lane (lr = co, lq), stores lane (px = lane % 16, co = lane / 16)
    assert _b128_worst(lambda lane: 40 * (lane & 15) + 4 * (lane >> 4)) == 1
    assert _b128_worst(lambda lane: 36 * (lane & 15) + 4 * (lane >> 4)) > 1
    assert _b32_store_worst(lambda lane: 40 * (lane >> 4) + (lane & 15)) == 2


def test_encoder_linear_forward_operand_reads():
    # chunk image rows of 72 dwords; lane (lr = row, lq) reads the float4 at 16 j + 4 lq
    for j in range(4):
        assert _b128_worst(lambda lane, j=j: 72 * (lane & 15) + 16 * j + 4 * (lane >> 4)) == 1
    assert _b128_worst(lambda lane: 68 * (lane & 15) + 4 * (lane >> 4)) > 1


def test_split_dgrad_cell_items_cover_the_band_once():
    # wave w takes items w, w + 8, w + 16 (< 18): item W = (pooled row W / 6, px half (W / 3) & 1, pair group W % 3), lane = (pair, px)
    seen = set()
    for wave in range(8):
        for j in range(3):
            W = wave + 8 * j
            if W >= 18:
                continue
            for lane in range(64):
                pr, px, pyl = 8 * (W % 3) + (lane & 7), 8 * ((W // 3) & 1) + ((lane >> 3) & 7), W // 6
                assert (pr, px, pyl) not in seen
                seen.add((pr, px, pyl))
    assert len(seen) == 24 * 16 * 3
