"""CPU replay of the LDS addressing of csrc/gemm256x.hip (round 4: the prior's GEMM tile on v_mfma_f32_16x16x32).

The LDS image is the one gemm256n.hip's LDS-DMA writes (16-byte chunks of a 128-byte row XOR-swizzled on the SOURCE address with
(row / 2) & 7); the new kernel reads (16 rows) x (32 k) fragments from it: lane l -> row l % 16 of the tile, chunk (4 s | l / 16).
Checked here, against the lane groups a ds_read_b128 is serviced in (MI355X_MICROARCH.md, LDS table):
  * every fragment read fetches the chunk it means to (the read-side XOR undoes the write-side one for every tile / sub-step),
  * every group of 16 lanes covers 16 distinct 16-byte slots of the 256-byte bank window: conflict free.
The constants mirror the kernel source (Cfg256X, rd0 / sw / lg in gemm256x_kernel, dch in the DMA geometry)."""

ROWB = 128
B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               [32 + x for x in list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28))],
               [32 + x for x in list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]]


def dma_image():
    """pos[row][slot] = source chunk the LDS-DMA of wave w, lane (rl, pch) puts at (row = w * 8 + rl within a 64-row piece, slot pch)."""
    img = {}
    for w in range(8):
        for lane in range(64):
            rl, pch = lane >> 3, lane & 7
            dch = pch ^ ((((w & 1) << 2) + (rl >> 1)) & 7)
            img[(w * 8 + rl, pch)] = dch
    return img


def read_offset(lane, s, tile):
    l15, lg = lane & 15, lane >> 4
    sw = (l15 >> 1) & 7
    rd0 = l15 * ROWB + ((lg ^ sw) << 4)
    return (rd0 ^ (s << 6)) + tile * 2048


def test_fragment_reads_fetch_the_chunk_they_mean():
    img = dma_image()
    for tile in range(4):
        for s in range(2):
            for lane in range(64):
                off = read_offset(lane, s, tile)
                row, slot = off // ROWB, (off % ROWB) // 16
                assert row == tile * 16 + (lane & 15)
                assert img[(row, slot)] == 4 * s + (lane >> 4), (tile, s, lane)


def test_fragment_reads_are_bank_conflict_free():
    for tile in range(4):
        for s in range(2):
            for grp in B128_GROUPS:
                slots = {(read_offset(l, s, tile) % 256) // 16 for l in grp}
                assert len(slots) == 16, (tile, s, grp, sorted(slots))
