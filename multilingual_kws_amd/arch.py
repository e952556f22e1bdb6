"""Algorithmic work per stage of the embedding forward (for roofline arithmetic).

Mirrors the architecture table of csrc/mkws_embed_arch.h.  flops = 2*MAC; bytes = fp32 input +
output (+ weights, gate, residual) that a launch must move if nothing is cached: the algorithmic
minimum for that launch, not measured traffic.
"""
BLOCKS = [  # name, in, out, kernel, stride, expand
    ("1a", 32, 16, 3, 1, 1),
    ("2a", 16, 24, 3, 2, 6), ("2b", 24, 24, 3, 1, 6),
    ("3a", 24, 40, 5, 2, 6), ("3b", 40, 40, 5, 1, 6),
    ("4a", 40, 80, 3, 2, 6), ("4b", 80, 80, 3, 1, 6), ("4c", 80, 80, 3, 1, 6),
    ("5a", 80, 112, 5, 1, 6), ("5b", 112, 112, 5, 1, 6), ("5c", 112, 112, 5, 1, 6),
    ("6a", 112, 192, 5, 2, 6), ("6b", 192, 192, 5, 1, 6), ("6c", 192, 192, 5, 1, 6), ("6d", 192, 192, 5, 1, 6),
    ("7a", 192, 320, 3, 1, 6),
]
FRONTEND_BYTES_PER_CLIP_F32 = 16000 * 4 + 49 * 40 * 4       # 71 840
FRONTEND_BYTES_PER_CLIP_I16 = 16000 * 2 + 49 * 40 * 4       # 39 840
EMBED_FLOPS_PER_CLIP = 2 * 32974496


def _down(h, w, k):
    pt, pl = k // 2 - (1 - h % 2), k // 2 - (1 - w % 2)
    return (h + pt + k // 2 - k) // 2 + 1, (w + pl + k // 2 - k) // 2 + 1


def stage_costs(B):
    """{stage name: (flops, bytes)} for one forward pass at batch B (stage names = profile/tap names)."""
    c = {}
    h, w = 25, 20
    c["stem"] = (2 * B * h * w * 9 * 32, 4 * (B * 49 * 40 + B * h * w * 32 + 9 * 32))
    # fused stem + block-1a depthwise launch: spectrogram in, depthwise output out
    c["stem_dw"] = (c["stem"][0] + 2 * B * h * w * 9 * 32, 4 * (B * 49 * 40 + B * h * w * 32 + 2 * 9 * 32 + B * 32))
    # stem + whole block 1a in one launch: spectrogram in, block-1a output [25,20,16] out
    c["stem_block1a"] = (c["stem_dw"][0] + 2 * B * 2 * 32 * 8 + 2 * B * h * w * 32 * 16,
                         4 * (B * 49 * 40 + B * h * w * 16 + 2 * 9 * 32 + 2 * 32 * 8 + 32 * 16))
    for name, cin, cout, k, s, e in BLOCKS:
        p = "block" + name
        ce, se = cin * e, max(1, int(cin * 0.25))
        m_in = B * h * w
        if e != 1:
            c[p + "_expand"] = (2 * m_in * cin * ce, 4 * (m_in * cin + m_in * ce + cin * ce))
        if s == 2:
            h, w = _down(h, w, k)
        m_out = B * h * w
        c[p + "_dw"] = (2 * m_out * k * k * ce, 4 * (m_in * ce + m_out * ce + k * k * ce + B * ce))
        if e != 1:   # fused expand + depthwise launch: reads the block input, writes the depthwise output
            c[p + "_front"] = (c[p + "_expand"][0] + c[p + "_dw"][0],
                               4 * (m_in * cin + cin * ce + m_out * ce + k * k * ce + B * ce))
        c[p + "_gate"] = (2 * B * 2 * ce * se, 4 * (2 * B * ce + 2 * ce * se))
        if e != 1:   # whole-block launch (tiny images): block input + output + every weight of the block, once
            wbytes = cin * ce + k * k * ce + 2 * ce * se + ce * cout
            c[p + "_block"] = (2 * (m_in * cin * ce + m_out * k * k * ce + B * 2 * ce * se + m_out * ce * cout),
                               4 * (m_in * cin + m_out * cout + (m_out * cout if (s == 1 and cin == cout) else 0) + wbytes))
        res = m_out * cout if (s == 1 and cin == cout) else 0
        c[p] = (2 * m_out * ce * cout, 4 * (m_out * ce + B * ce + m_out * cout + res + ce * cout))
    m = B * h * w
    c["top"] = (2 * m * 320 * 1280, 4 * (m * 320 + m * 1280 + 320 * 1280))
    c["gap"] = (m * 1280, 4 * (m * 1280 + B * 1280))
    c["dense"] = (2 * B * 1280 * 2048, 4 * (B * 1280 + B * 2048 + 1280 * 2048))
    c["dense_1"] = (2 * B * 2048 * 2048, 4 * (B * 2048 * 2 + 2048 * 2048))
    c["dense_2"] = (2 * B * 2048 * 1024, 4 * (B * 2048 + B * 1024 + 2048 * 1024))
    return c
