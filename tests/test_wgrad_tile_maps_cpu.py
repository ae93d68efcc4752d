"""CPU: the index maps of csrc/conv_wgrad_tile.hip restated in Python and checked for the properties the kernels rely on (no GPU needed;
the kernels themselves are compared with autograd in tests/test_gpu_wgrad_tile.py):
  * block_of(): every (gradient block, pixel split) pair gets exactly one workgroup id (the XCD-aware orders of round 4 measured neutral
    and were removed in round 5);
  * the LDS-DMA slot tables of wgrad_tile_dma_kernel: a tile's dy / x area is a sequence of 1 KB pieces, lane l of piece p fills the
    16-byte slot at byte p*1024 + l*16 of the area; slot -> (pixel, 16-byte channel segment) with the padded pixel pitch must reach every
    (pixel, segment) of the tile exactly once, everything else (padding, the round-up of the area) must be a zero-page fetch;
  * the ring bookkeeping of the loader waves: the counted vmcnt wait in front of barrier i leaves exactly the pieces of the younger
    tiles in flight and the stage a new tile overwrites is the one the MFMA waves finished one barrier earlier."""
import itertools

import pytest


def block_of(wid, out_tiles, ksplit):
    return wid % out_tiles, wid // out_tiles


@pytest.mark.parametrize('out_tiles', [1, 2, 3, 4, 6, 8, 16, 32])
def test_block_of_is_a_bijection(out_tiles):
    for ks in (1, 2, 3, 5, 7, 24, 128):
        seen = set()
        for wid in range(out_tiles * ks):
            b, split = block_of(wid, out_tiles, ks)
            assert 0 <= b < out_tiles and 0 <= split < ks and (b, split) not in seen
            seen.add((b, split))
        assert len(seen) == out_tiles * ks


def slot_table(n_pieces, pitch, npix, nseg):
    """slot -> (pixel, segment) or None, as the loaders compute it: so = piece*1024 + lane*16; pix = so // pitch; seg = (so % pitch) // 16"""
    out = []
    for piece, lane in itertools.product(range(n_pieces), range(64)):
        so = piece * 1024 + lane * 16
        pix, seg = so // pitch, (so % pitch) // 16
        out.append((pix, seg) if pix < npix and seg < nseg else None)
    return out


@pytest.mark.parametrize('cof,cif,th,stride,ext', [(4, 4, 2, 1, 0), (2, 2, 3, 1, 2), (2, 2, 1, 2, 2), (1, 2, 4, 1, 0), (2, 1, 2, 1, 4), (4, 1, 5, 1, 0)])
def test_dma_slot_tables_cover_every_tile_vector_exactly_once(cof, cif, th, stride, ext):
    tw = 32
    pd, px = 32 * cof * 2 + 32, 32 * cif * 2 + (32 if stride == 1 else 16)       # padded pixel pitches (bytes) of the dy / x area
    hw, hh = (tw - 1) * stride + 1 + ext, (th - 1) * stride + 1 + ext
    for pitch, npix, nseg in ((pd, th * tw, 4 * cof), (px, hh * hw, 4 * cif)):
        area = -(-npix * pitch // 1024) * 1024                                  # rounded up to whole pieces
        tab = slot_table(area // 1024, pitch, npix, nseg)
        real = [t for t in tab if t is not None]
        assert len(real) == len(set(real)) == npix * nseg                          # each (pixel, segment) exactly once
        assert pitch % 32 == 0 and (pitch // 32) % 2 == 1 or stride == 2           # odd 32-byte units per pixel step (transpose reads)
        # the slot of (pixel, segment) is where the MFMA waves read it: byte pixel*pitch + segment*16
        for i, t in enumerate(tab):
            if t is not None:
                assert i * 16 == t[0] * pitch + t[1] * 16


@pytest.mark.parametrize('nst,ntl', [(3, 1), (3, 2), (3, 7), (4, 1), (4, 3), (4, 9)])
def test_ring_waits_and_stage_reuse(nst, ntl):
    ppw = 5                                             # pieces a loader wave issues per tile
    issued, landed_upto = [], -1                        # tiles in issue order; highest tile whose pieces the counted wait guarantees
    stage_of, busy = {}, {}                             # tile -> stage; stage -> tile the MFMA waves may still be reading
    pro = min(ntl, nst - 1)
    for j in range(pro):
        issued.append(j); stage_of[j] = j
    stg = pro
    for i in range(ntl):
        younger = min(ntl - 1 - i, nst - 2)
        outstanding_allowed = younger * ppw             # s_waitcnt vmcnt(N): in-order completion -> everything but the last N pieces landed
        total = len(issued) * ppw
        landed_upto = (total - outstanding_allowed) // ppw - 1
        assert landed_upto >= i and issued[:landed_upto + 1] == list(range(landed_upto + 1))      # tile i has landed at barrier i
        # barrier i: MFMA waves are done with tile i-1 and start tile i
        busy = {stage_of[i]: i}
        nx = i + nst - 1
        if nx < ntl:
            assert stg not in busy and stage_of.get(i - 1, stg) == stg or i == 0 and stg == nst - 1      # overwrites the stage of tile i-1
            issued.append(nx); stage_of[nx] = stg
            stg = 0 if stg + 1 == nst else stg + 1
    assert issued == list(range(ntl)) and all(stage_of[t] == t % nst for t in range(ntl))
