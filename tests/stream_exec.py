"""Execute a tools/synth_stream.py op list two ways: on the CPU oracle strictly in decode order, and through the GPU
ctx layer (which reorders into phases).  Shared by the ctx parity test and the frame-pipeline bench."""
import numpy as np

from openhevc_amd import lib as L


def chroma_dims(W, H):
    return [(H, W), (H // 2, W // 2), (H // 2, W // 2)]


def mc_params(op, c_idx):
    """(x, y, w, h, [(sx, sy, mx, my) per ref]) in plane samples -- luma_mc_*/chroma_mc_* (hevc.c:1641-1949), 4:2:0."""
    sh = 1 if c_idx else 0
    x, y, w, h = op["x0"] >> sh, op["y0"] >> sh, op["w"] >> sh, op["h"] >> sh
    refs = []
    for (mvx, mvy) in op["mv"]:
        refs.append((x + (mvx >> (2 + sh)), y + (mvy >> (2 + sh)), mvx & (7 if sh else 3), mvy & (7 if sh else 3)))
    return x, y, w, h, refs


def window(ref_plane, sx, sy, w, h, taps):
    before, after = (3, 4) if taps == 8 else (1, 2)
    rows = np.clip(np.arange(sy - before, sy + h + after), 0, ref_plane.shape[0] - 1)
    cols = np.clip(np.arange(sx - before, sx + w + after), 0, ref_plane.shape[1] - 1)
    return np.ascontiguousarray(ref_plane[rows][:, cols]), before


def run_oracle(oracle, po, bd, W, H, cur, refs, ops, fops):
    """cur: list of 3 planes (modified in place and returned), refs: list of [3 planes]."""
    for op in ops:
        if op["t"] == "mc":
            for c_idx in range(3):
                x, y, w, h, rp = mc_params(op, c_idx)
                luma = c_idx == 0
                kw = dict(denom=op["denom"], wx0=op["wx"][0], wx1=op["wx"][1], ox0=op["ox"][0], ox1=op["ox"][1])
                win0, b = window(refs[op["ref"][0]][c_idx], rp[0][0], rp[0][1], w, h, 8 if luma else 4)
                if not op["bi"]:
                    blk = oracle.mc(bd, luma, po.MC_UNI_W if op["weighted"] else po.MC_UNI, win0, b, b, w, h, rp[0][2], rp[0][3], **kw)
                else:
                    tmp = oracle.mc(bd, luma, po.MC_PUT, win0, b, b, w, h, rp[0][2], rp[0][3])
                    src2 = np.zeros((h, 64), np.int16); src2[:, :w] = tmp
                    win1, b = window(refs[op["ref"][1]][c_idx], rp[1][0], rp[1][1], w, h, 8 if luma else 4)
                    blk = oracle.mc(bd, luma, po.MC_BI_W if op["weighted"] else po.MC_BI, win1, b, b, w, h, rp[1][2], rp[1][3], src2=src2, **kw)
                cur[c_idx][y:y + h, x:x + w] = blk
        elif op["t"] == "tu":
            sh = 1 if op["c_idx"] else 0
            xy = np.array([[op["x0"] >> sh, op["y0"] >> sh]], np.int32)
            oracle.tu_batch(bd, op["kind"], op["log2"], op["coeffs"][None], cur[op["c_idx"]], xy)
        elif op["t"] == "intra":
            oracle.intra_pred(bd, cur, W, H, op["x0"], op["y0"], op["log2"], op["c_idx"], op["mode"], op["cands"],
                              chroma_format_idc=1, strong=1, smoothing_disabled=0, log2_ctb_size=6, log2_min_tb_size=2)
        elif op["t"] == "pcm":          # put_pcm: dst = sample << (BIT_DEPTH - pcm_bit_depth), hevcdsp_template.c:30-43
            sh = 1 if op["c_idx"] else 0
            n = 1 << op["log2"]
            x, y = op["x0"] >> sh, op["y0"] >> sh
            cur[op["c_idx"]][y:y + n, x:x + n] = (op["samples"] << (bd - op["pcm_bd"])).astype(cur[op["c_idx"]].dtype)
    for vertical in (1, 0):
        for op in fops:
            if op["t"] == "dbk" and op["vertical"] == vertical:
                if op["c_idx"] == 0:
                    oracle.deblock_luma(bd, vertical, cur[0], op["x"], op["y"], op["beta"], op["tc"], op["no_p"], op["no_q"])
                else:
                    oracle.deblock_chroma(bd, vertical, cur[op["c_idx"]], op["x"], op["y"], op["tc"], op["no_p"], op["no_q"])
    if any(op["t"] == "sao" for op in fops):
        src = [np.ascontiguousarray(np.pad(p, 1, mode="edge")) for p in cur]     # deblocked copy (+ ring standing in for frame padding)
        for op in fops:
            if op["t"] != "sao":
                continue
            c = op["c_idx"]
            dst = np.ascontiguousarray(np.pad(cur[c], 1, mode="edge"))
            if op["band"]:
                oracle.sao_band(bd, dst, src[c], op["x"] + 1, op["y"] + 1, op["w"], op["h"], op["offset_val"], op["klass"])
            else:
                oracle.sao_edge(bd, 0, dst, src[c], op["x"] + 1, op["y"] + 1, op["w"], op["h"], op["offset_val"], op["klass"], op["borders"])
            cur[c][op["y"]:op["y"] + op["h"], op["x"]:op["x"] + op["w"]] = dst[op["y"] + 1:op["y"] + 1 + op["h"], op["x"] + 1:op["x"] + 1 + op["w"]]
    return cur


def pcm_samples(op, bd=None):
    """Samples as the host would hand them over: already << (bit_depth - pcm_bit_depth)."""
    return (op["samples"] << (op["bd"] - op["pcm_bd"])).astype(np.int16)


def record_gpu(ctx, W, H, ref_slots, ops, fops):
    """Record the op list into an ohevc ctx (frame_begin must have been called)."""
    geom = L.IntraGeom(W, H, 1, 6, 2, 1, 0, 0)
    for op in ops:
        if op["t"] == "mc":
            for c_idx in range(3):
                x, y, w, h, rp = mc_params(op, c_idx)
                j = np.zeros(1, L.MC_JOB)
                j["x"], j["y"], j["w"], j["h"], j["plane"] = x, y, w, h, c_idx
                j["flags"] = (L.MC_BI if op["bi"] else 0) | (L.MC_WEIGHTED if op["weighted"] else 0)
                for s in (0, 1):
                    j[f"sx{s}"], j[f"sy{s}"], j[f"mx{s}"], j[f"my{s}"] = np.clip(rp[s][0], -32768, 32767), np.clip(rp[s][1], -32768, 32767), rp[s][2], rp[s][3]
                    j[f"ref{s}"] = ref_slots[op["ref"][s]]
                    j[f"wx{s}"], j[f"ox{s}"] = op["wx"][s], op["ox"][s]
                j["denom"] = op["denom"]
                ctx.rec_mc(j)
        elif op["t"] == "tu":
            sh = 1 if op["c_idx"] else 0
            ctx.rec_tu(op["c_idx"], op["x0"] >> sh, op["y0"] >> sh, op["log2"], op["kind"], op["coeffs"], op["intra"])
        elif op["t"] == "intra":
            ctx.rec_intra(L.intra_make_job(geom, op["x0"], op["y0"], op["log2"], op["c_idx"], op["mode"], op["cands"]))
        elif op["t"] == "pcm":
            sh = 1 if op["c_idx"] else 0
            ctx.rec_tu(op["c_idx"], op["x0"] >> sh, op["y0"] >> sh, op["log2"], L.TU_PCM, pcm_samples(op), 1)
    for op in fops:
        if op["t"] == "dbk":
            j = np.zeros(1, L.DBK_JOB)
            j["x"], j["y"], j["plane"], j["beta"], j["tc"] = op["x"], op["y"], op["c_idx"], op["beta"], op["tc"]
            j["flags"] = (L.DBK_VERTICAL_EDGE if op["vertical"] else 0) | (L.DBK_NO_P0 * op["no_p"][0]) | (L.DBK_NO_P1 * op["no_p"][1]) | \
                         (L.DBK_NO_Q0 * op["no_q"][0]) | (L.DBK_NO_Q1 * op["no_q"][1])
            ctx.rec_deblock(j)
        else:
            j = np.zeros(1, L.SAO_JOB)
            j["x"], j["y"], j["w"], j["h"], j["plane"] = op["x"], op["y"], op["w"], op["h"], op["c_idx"]
            j["type"] = L.SAO_BAND if op["band"] else L.SAO_EDGE
            j["klass"] = op["klass"]
            b = op["borders"]
            j["borders"] = b[0] | (b[1] << 1) | (b[2] << 2) | (b[3] << 3)
            j["offset_val"] = op["offset_val"]
            ctx.rec_sao(j)


def ops_to_arrays(W, H, ref_slots, ops, fops):
    """Convert an op list into the numpy job arrays ohevc_rec_*_bulk take (intra jobs go through ohevc_intra_make_job)."""
    geom = L.IntraGeom(W, H, 1, 6, 2, 1, 0, 0)
    mc, intra, desc, coeffs, dbk, sao = [], [], [], [], [], []
    for op in ops:
        if op["t"] == "mc":
            for c_idx in range(3):
                x, y, w, h, rp = mc_params(op, c_idx)
                j = np.zeros(1, L.MC_JOB)[0]
                j["x"], j["y"], j["w"], j["h"], j["plane"] = x, y, w, h, c_idx
                j["flags"] = (L.MC_BI if op["bi"] else 0) | (L.MC_WEIGHTED if op["weighted"] else 0)
                for s in (0, 1):
                    j[f"sx{s}"], j[f"sy{s}"] = np.clip(rp[s][0], -32768, 32767), np.clip(rp[s][1], -32768, 32767)
                    j[f"mx{s}"], j[f"my{s}"] = rp[s][2], rp[s][3]
                    j[f"ref{s}"] = ref_slots[op["ref"][s]]
                    j[f"wx{s}"], j[f"ox{s}"] = op["wx"][s], op["ox"][s]
                j["denom"] = op["denom"]
                mc.append(j)
        elif op["t"] == "tu":
            sh = 1 if op["c_idx"] else 0
            desc.append([op["c_idx"], op["x0"] >> sh, op["y0"] >> sh, op["log2"], op["kind"], op["intra"]])
            coeffs.append(np.ascontiguousarray(op["coeffs"], dtype=np.int16).reshape(-1))
        elif op["t"] == "pcm":
            sh = 1 if op["c_idx"] else 0
            desc.append([op["c_idx"], op["x0"] >> sh, op["y0"] >> sh, op["log2"], L.TU_PCM, 1])
            coeffs.append(pcm_samples(op).reshape(-1))
        else:
            intra.append(L.intra_make_job(geom, op["x0"], op["y0"], op["log2"], op["c_idx"], op["mode"], op["cands"])[0])
    for op in fops:
        if op["t"] == "dbk":
            j = np.zeros(1, L.DBK_JOB)[0]
            j["x"], j["y"], j["plane"], j["beta"], j["tc"] = op["x"], op["y"], op["c_idx"], op["beta"], op["tc"]
            j["flags"] = (L.DBK_VERTICAL_EDGE if op["vertical"] else 0) | (L.DBK_NO_P0 * op["no_p"][0]) | (L.DBK_NO_P1 * op["no_p"][1]) | \
                         (L.DBK_NO_Q0 * op["no_q"][0]) | (L.DBK_NO_Q1 * op["no_q"][1])
            dbk.append(j)
        else:
            j = np.zeros(1, L.SAO_JOB)[0]
            j["x"], j["y"], j["w"], j["h"], j["plane"] = op["x"], op["y"], op["w"], op["h"], op["c_idx"]
            j["type"] = L.SAO_BAND if op["band"] else L.SAO_EDGE
            j["klass"] = op["klass"]
            b = op["borders"]
            j["borders"] = b[0] | (b[1] << 1) | (b[2] << 2) | (b[3] << 3)
            j["offset_val"] = op["offset_val"]
            sao.append(j)
    return dict(mc=np.array(mc, dtype=L.MC_JOB), intra=np.array(intra, dtype=L.INTRA_JOB),
                tu_desc=np.array(desc, dtype=np.int32).reshape(-1, 6),
                tu_coeffs=np.concatenate(coeffs) if coeffs else np.zeros(0, np.int16),
                dbk=np.array(dbk, dtype=L.DBK_JOB), sao=np.array(sao, dtype=L.SAO_JOB))


# ---------------------------------------------------------------- table-driver encoding (oracle/table_driver.c)
OP_WORDS = 24


def encode_driver_ops(ops, fops, pcm_ops=()):
    """Flatten an op list into the int32[n,24] + coefficient + pcm-bit arrays ohref_drive_tables takes."""
    rows, coeffs, bits = [], [], bytearray()
    coff = 0

    def pack_pcm(op):
        acc, nb = 0, 0
        start = len(bits)
        for v in np.asarray(op["samples"]).reshape(-1):
            acc = (acc << op["pcm_bd"]) | int(v); nb += op["pcm_bd"]
            while nb >= 8:
                bits.append((acc >> (nb - 8)) & 255); nb -= 8
                acc &= (1 << nb) - 1
        if nb:
            bits.append((acc << (8 - nb)) & 255)
        bits.extend(b"\0" * 8)                      # get_bits reads ahead
        r = [5, op["c_idx"], op["x"], op["y"], op["log2"], op["pcm_bd"], start, len(bits) - start]
        rows.append(r + [0] * (OP_WORDS - len(r)))

    for op in ops:
        if op["t"] == "mc":
            for c_idx in range(3):
                x, y, w, h, rp = mc_params(op, c_idx)
                r = [0, c_idx, x, y, w, h, (1 if op["bi"] else 0) | (2 if op["weighted"] else 0), op["ref"][0], op["ref"][1],
                     rp[0][0], rp[0][1], rp[0][2], rp[0][3], rp[1][0], rp[1][1], rp[1][2], rp[1][3],
                     op["denom"], op["wx"][0], op["wx"][1], op["ox"][0], op["ox"][1]]
                rows.append(r + [0] * (OP_WORDS - len(r)))
        elif op["t"] == "tu":
            sh = 1 if op["c_idx"] else 0
            r = [1, op["c_idx"], op["x0"] >> sh, op["y0"] >> sh, op["log2"], op["kind"], coff]
            c = np.ascontiguousarray(op["coeffs"], dtype=np.int16).reshape(-1)
            coeffs.append(c); coff += c.size
            rows.append(r + [0] * (OP_WORDS - len(r)))
        elif op["t"] == "intra":
            r = [2, op["c_idx"], op["x0"], op["y0"], op["log2"], op["mode"]] + list(op["cands"])
            rows.append(r + [0] * (OP_WORDS - len(r)))
        else:                   # pcm: pack the samples MSB-first, pcm_bd bits each, like the bitstream carries them
            sh = 1 if op["c_idx"] else 0
            pack_pcm(dict(c_idx=op["c_idx"], x=op["x0"] >> sh, y=op["y0"] >> sh, log2=op["log2"], pcm_bd=op["pcm_bd"], samples=op["samples"]))
    for op in pcm_ops:      # dict(c_idx, x, y, log2, pcm_bd, samples[N,N])
        pack_pcm(op)
    for op in ():
        n = 1 << op["log2"]
        acc, nb = 0, 0
        start = len(bits)
        for v in np.asarray(op["samples"]).reshape(-1):
            acc = (acc << op["pcm_bd"]) | int(v); nb += op["pcm_bd"]
            while nb >= 8:
                bits.append((acc >> (nb - 8)) & 255); nb -= 8
        if nb:
            bits.append((acc << (8 - nb)) & 255)
        bits.extend(b"\0" * 8)                       # get_bits reads ahead
        r = [5, op["c_idx"], op["x"], op["y"], op["log2"], op["pcm_bd"], start, len(bits) - start]
        rows.append(r + [0] * (OP_WORDS - len(r)))
    for op in fops:
        if op["t"] == "dbk":
            r = [3, op["c_idx"], op["x"], op["y"], op["vertical"], op["beta"], op["tc"][0], op["tc"][1],
                 op["no_p"][0], op["no_p"][1], op["no_q"][0], op["no_q"][1]]
        else:
            r = [4, op["c_idx"], op["x"], op["y"], op["w"], op["h"], int(op["band"]), op["klass"]] + list(op["borders"]) + list(op["offset_val"])
        rows.append(r + [0] * (OP_WORDS - len(r)))
    return (np.array(rows, dtype=np.int32).reshape(-1, OP_WORDS),
            np.concatenate(coeffs) if coeffs else np.zeros(8, np.int16), np.frombuffer(bytes(bits) + b"\0" * 16, dtype=np.uint8).copy())


def drive_tables(reflib, bd, W, H, cur, refs, enc, hevcdsp_hook=None, videodsp_hook=None, intra_hook=None, geom=None):
    """Run oracle/table_driver.c on host planes (numpy, modified in place).  reflib = ctypes CDLL of libhevcref.so."""
    import ctypes as C

    class DrvPic(C.Structure):
        _fields_ = [("data", C.c_void_p * 3), ("linesize", C.c_int32 * 3)]

    def pic(planes):
        p = DrvPic()
        for i, a in enumerate(planes):
            p.data[i] = a.ctypes.data; p.linesize[i] = a.strides[0]
        return p
    cp = pic(cur)
    rarr = (DrvPic * len(refs))(*[pic(r) for r in refs])
    ops, coeffs, bits = enc
    f = reflib.ohref_drive_tables
    f.restype = C.c_int
    rc = f(C.c_int(bd), C.c_int(W), C.c_int(H), C.byref(cp), rarr, C.c_int(len(refs)),
           ops.ctypes.data_as(C.c_void_p), C.c_int(len(ops)), coeffs.ctypes.data_as(C.c_void_p), bits.ctypes.data_as(C.c_void_p),
           C.c_void_p(hevcdsp_hook), C.c_void_p(videodsp_hook), C.c_void_p(intra_hook), C.c_void_p(geom))
    return rc


def check_switches(kind, product, names, threads=1, combos=((1, 2), (96, 1), (1, 0))):
    """Round-5 switches of the product library that no small stream trips on its own, forced: (a) every frame with intra levels goes to the
    context's long-chain stream (a picture of the golden streams has a few dozen levels, the default threshold is 96) - the hand-over between a
    context's two streams at every picture; (b) the three forms coefficients cross the bus in - the non-zero 4x4 groups (2, the default), the col_limit rectangle (1, round 5), whole (0, rounds 1-4).  Same pictures."""
    import ctypes as C
    from oracle import pystream as ps
    from test_stream_cpu import frames_md5, load_golden
    product.ohevc_debug_set_long_chain_levels.argtypes = [C.c_int]
    product.ohevc_debug_set_compact_coeffs.argtypes = [C.c_int]
    if threads > 1 and ps.have("c"):
        # the reference's own frame threads do not reproduce its single-thread pictures on every stream (cross_444_10b_tqb: 3 runs of 3 differ, on
        # its C and on its SSE tables alike): such a stream says nothing about the back end
        names = [n for n in names if frames_md5(ps.decode_stream("c", load_golden(n)[0], threads, 1)) == load_golden(n)[1]]
    elif threads > 1:
        names = [n for n in names if n != "cross_444_10b_tqb"]
    try:
        for levels, compact in combos:
            product.ohevc_debug_set_long_chain_levels(levels)
            product.ohevc_debug_set_compact_coeffs(compact)
            for name in names:
                aus, md5 = load_golden(name)
                assert frames_md5(ps.decode_stream(kind, aus, threads, 1)) == md5, f"{name}: long_chain_levels {levels}, compact {compact}, {threads} thread(s)"
    finally:
        product.ohevc_debug_set_long_chain_levels(96)
        product.ohevc_debug_set_compact_coeffs(2)
