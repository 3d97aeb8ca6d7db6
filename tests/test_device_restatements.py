"""The tile image the LDS-DMA staging of the tiled sweeps writes (csrc/glrm_tiled.hpp: dma_tile_all) restated on the host: one
global_load_lds_dwordx4 lane per 16-byte piece of the image, wave w of NW taking the KiB blocks w, w + NW, ...; padded rows compute the
source of each piece from its position (the pad piece re-reads its neighbour), ROT rows are a plain copy.  The image must equal what the
load / ds_write staging of rounds 1-2 (stage_tile) produced: row r of the tile at byte r * stride, its kp * 8 bytes unchanged."""
import numpy as np
import pytest


def dma_image(src_rows, G, R, NW, rot):
    kpb = G * R * 8
    stride = kpb if rot else kpb + 16
    cpr = stride // 16
    rows = src_rows.shape[0]
    total = rows * cpr
    src = src_rows.reshape(-1).view(np.uint8)
    lds = np.full(rows * stride + 1024, 0xEE, dtype=np.uint8)   # canary behind the image: masked lanes must not write
    written = np.zeros(rows * stride + 1024, dtype=np.int32)
    for wave in range(NW):
        base = wave * 64
        while base < total:                                      # for (base = wave * 64; base < total; base += NW * 64)
            for lane in range(64):
                c = base + lane
                if c >= total:
                    continue                                     # masked lane
                if rot:
                    s = c * 16
                else:
                    row, cc = divmod(c, cpr)
                    cc = min(cc, cpr - 2)                        # the pad piece re-reads the last data piece
                    s = row * kpb + cc * 16
                d = base * 16 + lane * 16                        # M0 base + lane * 16
                lds[d:d + 16] = src[s:s + 16]
                written[d:d + 16] += 1
            base += NW * 64
    return lds, written, stride


@pytest.mark.parametrize("G,R", [(4, 2), (4, 4), (4, 8), (8, 8), (16, 8)])
@pytest.mark.parametrize("rows", [1, 5, 63, 288, 560])
@pytest.mark.parametrize("rot", [False, True])
def test_dma_staged_tile_equals_the_staged_rows(G, R, rows, rot):
    kp = G * R
    rng = np.random.default_rng(G * 1000 + R * 10 + rows)
    src = rng.standard_normal((rows, kp))
    lds, written, stride = dma_image(src, G, R, 16, rot)
    for r in range(rows):
        got = lds[r * stride: r * stride + kp * 8].view(np.float64)
        assert np.array_equal(got, src[r]), (r,)
    assert written[:rows * stride].max() == 1 and written[:rows * stride].min() == 1   # every byte of the image exactly once
    assert written[rows * stride:].sum() == 0 and np.all(lds[rows * stride:] == 0xEE)  # nothing behind the image


# ------------------------------------------------------------------------------------------------ slot_prefix (csrc/glrm_multi.hpp)
def slot_prefix_min(v, lg):
    """The inclusive prefix minimum over the lanes of a slot as the general sweeps form it for MultinomialOrdinalLoss's thresholds: slots of
    <= 16 lanes by row_shr:1/2/4/8 DPP steps (a DPP row is 16 lanes; a lane whose source falls outside its row keeps an undefined value, which
    the `sub >= o` select discards), wider slots by __shfl_up over the wave."""
    v = np.array(v, dtype=float)
    P = 1 << lg
    sub = np.arange(64) & (P - 1)
    offsets = [o for o in (1, 2, 4, 8) if o < P] if lg <= 4 else [1 << i for i in range(lg)]
    if lg <= 4 and P >= 2 and 2 not in offsets:
        offsets = [1, 2]            # the device code always runs the first two steps (lgP >= 2 in the engine)
    for o in offsets:
        t = np.full(64, np.nan)     # NaN = "undefined": must never be selected
        for lane in range(64):
            srcl = lane - o
            if lg <= 4:
                if srcl >= 0 and srcl // 16 == lane // 16:
                    t[lane] = v[srcl]
            else:
                t[lane] = v[srcl] if srcl >= 0 else v[lane]
        take = sub >= o
        assert not np.any(np.isnan(t[take])), "a selected lane read outside its DPP row"
        v = np.where(take & (t < v), t, v)
    return v


@pytest.mark.parametrize("lg", [2, 3, 4, 5, 6])
def test_slot_prefix_minimum_equals_the_running_minimum(lg):
    rng = np.random.default_rng(lg)
    P = 1 << lg
    for _ in range(50):
        v = rng.standard_normal(64)
        v[rng.random(64) < 0.2] = np.inf     # lanes outside the embedding / NaN thresholds enter as +Inf
        got = slot_prefix_min(v, lg)
        want = np.concatenate([np.minimum.accumulate(v[s:s + P]) for s in range(0, 64, P)])
        assert np.array_equal(got, want)
