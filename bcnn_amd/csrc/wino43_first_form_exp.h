// wino43_first_form_exp.h -- EXPERIMENT ONLY (-DBCNN_HIP_EXPERIMENT: libbcnn_hip_exp.so): the FIRST form of the F(4x4, 3x3)
// forward / dX kernel (round 5: 32 channels x 32 tiles per unit on v_mfma_f32_32x32x2_f32, twelve waves, M through LDS). The
// product library runs the second form (conv_winograd43b.hip) on every plane of whole 4 x 4 tiles, which is all the product
// rule admits; this one stays for planes that are NOT whole tiles (the RAG instantiations: 14 x 14, 7 x 7, odd sizes; forced
// with BCNN_HIP_WINOGRAD43=1, tests/test_winograd43.py) and as the A/B partner (BCNN_HIP_W43_FORM=1). DESIGN.md section 4.9.
// Included by conv_winograd43.hip inside namespace bcnn_hip.
#pragma once
// Timing experiments (tools/exp/variant.sh NAME conv_winograd43 "-DW43_ABL_..."; normal builds define none of them; results
// are wrong): W43_ABL_NOXFORM (no input transform arithmetic / V writes), NOEPI (no output transform, stores, statistics),
// NOMFMA, NODMA (U slab only for chunk 0), NOBAR (no barrier in the K loop), NOLDSRD (no fragment reads), NOLOAD (patch requests
// only for chunk 0), HITLOAD (the patch requests against one cache-resident kilobyte). DESIGN.md section 4.9 has the table.

struct Wino43Args {
    const float* src;  // x (forward) or dy (dX): [N][J][H][W]
    const float* upk;  // transformed weights [36][Jpad][Mpad], zero padded
    float* dst;        // [N][M][H][W]
    float* stats;      // optional: [M][tblocks][2]
    int N, J, M, H, W, TH, TW;
    unsigned T;
    int Jpad, Mpad, mblocks, tblocks, nunits;
    unsigned src_bytes, upk_bytes, dst_bytes, stats_bytes;
    // K-split tail: units [nunits, nunits + tail_units) -- the blocks a last, partly filled round would hold -- are not run as
    // units. Their chunks, flattened (unit-major), are dealt out evenly: workgroup i takes chunks [i * tail_q, (i + 1) * tail_q)
    // as one or two pieces and writes each piece's raw outputs to tail_scr[2 * i + piece][channel 0..31][tile 0..31][16].
    float* tail_scr;
    int tail_q, tail_units;
    unsigned tail_scr_bytes;
};

// the first five steps of wave_sum_dpp: lanes 31 / 63 end up with their half-wave's sum
__device__ __forceinline__ float w43_half_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
    return v;
}


// RAG: planes that are not whole 4 x 4 tiles (14 x 14, 7 x 7): the last tile row / column hangs over -- rows beyond H arrive
// as zeros through the row test, columns beyond W (which a 16-byte row load takes from the NEXT image row) are zeroed by
// selects, outputs beyond the plane are dropped by address and left out of the statistics; rows are 4-byte aligned only.
template <bool STATS, bool RAG>
__global__ __launch_bounds__(64 * W4_NW, 3) void wino43_kernel(const Wino43Args a) {
    __shared__ __attribute__((aligned(1024))) float lds[2 * W4_STAGE];  // 147,456 bytes: two stages; the epilogue's M in the second
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int HW = a.H * a.W;
    const bool xform = wid < 4;  // the transforming waves: channel 2 wid + lhi of the chunk, tile l31
    const rsrc_i4 rs_src = make_rsrc(a.src, a.src_bytes);
    const rsrc_i4 rs_u = make_rsrc(a.upk, a.upk_bytes);
    const rsrc_i4 rs_dst = make_rsrc(a.dst, a.dst_bytes);
    const rsrc_i4 rs_stats = make_rsrc(a.stats, STATS ? a.stats_bytes : 0u);
    const rsrc_i4 rs_scr = make_rsrc(a.tail_scr, a.tail_scr_bytes);
    const unsigned lds0 = lds_offset(&lds[0]);
    // LDS-DMA of U: 8 rows (k) x 32 floats per instruction; lane -> row lane / 8, floats 4 * (lane % 8) ..
    const unsigned u_voff = ((unsigned)(lane >> 3) * (unsigned)a.Mpad + (unsigned)(lane & 7) * 4u) * 4u;
    const unsigned per_img = (unsigned)(a.TH * a.TW);
    const int nch = a.Jpad / W4_KC;
    const unsigned row_bytes = (unsigned)a.W * 4u;

    // ---- this workgroup's work list: its whole units, then (K-split tail) one or two PIECES -- a chunk range of a tail unit
    // whose raw 4 x 4 outputs (a partial sum over its input channels: the output transform is linear) go to a scratch slot;
    // wino43_tail_fixup_kernel adds a unit's pieces in channel order. See wino_fused_kernel (conv_winograd_fused.hip). ----
    const int nreg = (int)blockIdx.x < a.nunits ? (a.nunits - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    int npieces = 0, pa_unit = 0, pa_k0 = 0, pa_n = 0, pb_n = 0;
    if (a.tail_units > 0) {
        const int c0 = (int)blockIdx.x * a.tail_q, c1 = min(c0 + a.tail_q, a.tail_units * nch);
        if (c0 < c1) {
            pa_unit = a.nunits + c0 / nch;
            pa_k0 = c0 % nch;
            pa_n = min(c1 - c0, nch - pa_k0);
            pb_n = (c1 - c0) - pa_n;  // > 0: the range runs on into the next unit
            npieces = pb_n > 0 ? 2 : 1;
        }
    }
    const int nitems = nreg + npieces;
    if (nitems == 0) return;

    // the item whose chunks are being REQUESTED (from the end of the previous item's K loop on, that is the next one)
    int kb = 0, nchunks = nch, slot = -1, m0 = 0, tb = 0, th = 0, ncols = 4, nrows = 4;
    bool tile_ok = false, pad_l = false, pad_r = false, edge = false;
    unsigned vbase = 0, edge_delta = 0, o00 = kOOB;
    unsigned v_top = kOOB, v_mid = kOOB, v_bot = kOOB, e_top = kOOB, e_mid = kOOB, e_bot = kOOB;  // !RAG: per-item row offsets
    auto start_item = [&](int it) {
        int unit;
        if (it < nreg) { unit = (int)blockIdx.x + it * (int)gridDim.x; kb = 0; nchunks = nch; slot = -1; }
        else if (it == nreg) { unit = pa_unit; kb = pa_k0; nchunks = pa_n; slot = 2 * (int)blockIdx.x; }
        else { unit = pa_unit + 1; kb = 0; nchunks = pb_n; slot = 2 * (int)blockIdx.x + 1; }
        const int mb = unit % a.mblocks;  // channel blocks of one tile block run together
        tb = unit / a.mblocks;
        m0 = mb * W4_BF;
        // this lane's tile (the same one for the input transform and for the output transform)
        const unsigned t = (unsigned)tb * W4_BT + (unsigned)l31;
        tile_ok = t < a.T;
        const unsigned n = tile_ok ? t / per_img : 0u;
        const unsigned rr = tile_ok ? t - n * per_img : 0u;
        th = (int)(rr / (unsigned)a.TW);
        const int tw = (int)(rr - (unsigned)th * (unsigned)a.TW);
        // byte offset of patch row 0 (image row 4 th - 1), own columns 4 tw .. 4 tw + 3, channel lhi of the wave's pair
        vbase = (n * (unsigned)a.J * (unsigned)HW + (unsigned)lhi * (unsigned)HW) * 4u +
                (unsigned)((4 * th - 1) * a.W + 4 * tw) * 4u;  // row -1 wraps: only used when that row exists
        pad_l = tw == 0; pad_r = tw + 1 == a.TW;
        const bool edge_l = l31 == 0 && !pad_l, edge_r = l31 == 31 && !pad_r;  // neighbour column not in a neighbouring lane
        edge = edge_l || edge_r;
        edge_delta = edge_l ? (unsigned)-4 : 16u;
        if (!RAG) {  // rows 1..4 of a whole tile always exist; rows 0 and 5 are padding at the top / bottom tile row.
            // The row step rides in the scalar offset (the range check sees the vector offset only), so a chunk's twelve
            // requests cost no address arithmetic in the transforming waves' serial section.
            const unsigned r1 = vbase + row_bytes;
            const bool top = tile_ok && th > 0, bot = tile_ok && 4 * th + 4 < a.H;
            v_mid = tile_ok ? r1 : kOOB;  v_top = top ? vbase : kOOB;  v_bot = bot ? r1 : kOOB;
            e_mid = tile_ok && edge ? r1 + edge_delta : kOOB;
            e_top = top && edge ? vbase + edge_delta : kOOB;
            e_bot = bot && edge ? r1 + edge_delta : kOOB;
        }
        o00 = tile_ok ? (n * (unsigned)a.M * (unsigned)HW + (unsigned)(4 * th * a.W + 4 * tw)) * 4u : kOOB;
        ncols = a.W - 4 * tw < 4 ? a.W - 4 * tw : 4;  // RAG: own columns / rows that exist
        nrows = a.H - 4 * th < 4 ? a.H - 4 * th : 4;
    };

    float p[6][4], e[6];  // a patch: own columns and the edge lanes' neighbour column (0.0 elsewhere)
    auto load_patch = [&](int kc) {
#ifdef W43_ABL_NOLOAD
        if (kc > 0) return;
#endif
        const unsigned soff = (unsigned)((kb + kc) * W4_KC + 2 * wid) * (unsigned)HW * 4u;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#if !defined(W43_ABL_HITLOAD) && !defined(W43_ABL_ROWADDR)   // ROWADDR: timing only, the per-chunk address arithmetic back
            if (!RAG) {
                const unsigned so = i == 0 ? soff : soff + (unsigned)(i - 1) * row_bytes;
                const buf_f32x4 v = buffer_load_f32x4(rs_src, (int)(i == 0 ? v_top : i == 5 ? v_bot : v_mid), (int)so, 0);
                p[i][0] = v[0]; p[i][1] = v[1]; p[i][2] = v[2]; p[i][3] = v[3];
                e[i] = buffer_load_f32(rs_src, (int)(i == 0 ? e_top : i == 5 ? e_bot : e_mid), (int)so, 0);
                continue;
            }
#endif
            const int ih = 4 * th - 1 + i;
            const bool row_ok = tile_ok && (unsigned)ih < (unsigned)a.H;
            const unsigned row = vbase + (unsigned)i * row_bytes;
#ifdef W43_ABL_HITLOAD   // timing only: the same instructions against one cache-resident kilobyte
            const buf_f32x4 v = buffer_load_f32x4(rs_src, (int)(row_ok ? (unsigned)lane * 16u + (unsigned)i * 1024u : kOOB), 0, 0);
#else
            const buf_f32x4 v = buffer_load_f32x4(rs_src, (int)(row_ok ? row : kOOB), (int)soff, 0);
#endif
            p[i][0] = v[0];
            p[i][1] = (RAG && ncols < 2) ? 0.f : v[1];
            p[i][2] = (RAG && ncols < 3) ? 0.f : v[2];
            p[i][3] = (RAG && ncols < 4) ? 0.f : v[3];
            e[i] = buffer_load_f32(rs_src, (int)((row_ok && edge) ? row + edge_delta : kOOB), (int)soff, 0);
        }
    };
    auto dma_u = [&](int kc, int stage) {  // this wave's positions: W4_PW x (8 rows of 32 floats)
#ifdef W43_ABL_NODMA
        if (kc > 0) return;
#endif
#pragma unroll
        for (int q = 0; q < W4_PW; ++q) {
            const int xi = W4_PW * wid + q;
            const unsigned soff = (((unsigned)xi * (unsigned)a.Jpad + (unsigned)((kb + kc) * W4_KC)) * (unsigned)a.Mpad + (unsigned)m0) * 4u;
            dma_row_x4(rs_u, lds0 + (unsigned)((stage * W4_STAGE + xi * W4_KC * 32) * 4), u_voff, soff);
        }
    };
    auto write_v = [&](int stage) {  // B^T d B -> V[xi][2 wid + lhi][l31]
#ifdef W43_ABL_NOXFORM
        {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i) sum += p[i][0] + p[i][1] + p[i][2] + p[i][3] + e[i];
            if (sum == 123.456f) lds[stage * W4_STAGE + W4_OP + lane] = sum;
            return;
        }
#endif
        float tt[6][6];  // columns first: tt[.][j] = B^T d[.][j]
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float col[6], out[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (j >= 1 && j <= 4) {
                    col[i] = p[i][j - 1];
                } else if (j == 0) {  // lane l - 1's last own column; the first lane of a half-wave fetched its own
                    const int ev = __builtin_bit_cast(int, e[i]);
                    const int l = __builtin_amdgcn_update_dpp(ev, __builtin_bit_cast(int, p[i][3]), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                    col[i] = pad_l ? 0.f : __builtin_bit_cast(float, l31 == 0 ? ev : l);
                } else {              // lane l + 1's first own column
                    const int ev = __builtin_bit_cast(int, e[i]);
                    const int r = __builtin_amdgcn_update_dpp(ev, __builtin_bit_cast(int, p[i][0]), 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
                    col[i] = pad_r ? 0.f : __builtin_bit_cast(float, l31 == 31 ? ev : r);
                }
            }
            w43_bt(col, out);
#pragma unroll
            for (int i = 0; i < 6; ++i) tt[i][j] = out[i];
        }
        float* v = lds + stage * W4_STAGE + W4_OP + (2 * wid + lhi) * 32 + l31;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float out[6];
            w43_bt(tt[i], out);
#pragma unroll
            for (int j = 0; j < 6; ++j) v[(6 * i + j) * W4_KC * 32] = out[j];
        }
    };

    // chunk 0 of the first item: U and V into stage 0, the patches of chunk 1 into the registers (every later item's chunk 0
    // is started underneath the previous item's epilogue, which keeps its M in stage 1 only)
    start_item(0);
    dma_u(0, 0);
    if (xform) {
        load_patch(0);
        write_v(0);
        if (nchunks > 1) load_patch(1);
    }
    for (int it = 0; it < nitems; ++it) {
        f32x16 acc[W4_PW];  // not cleared: the first k-step of chunk 0 multiplies onto a literal zero
        for (int kc = 0; kc < nchunks; ++kc) {
            const int cur = kc & 1;
            // U(kc) is older than the 12 patch requests of chunk kc + 1, which may fly on
            if (xform && kc + 1 < nchunks) dma_wait_n<12>(); else dma_wait();
#ifndef W43_ABL_NOBAR
            lds_barrier();  // stage cur holds chunk kc;
#endif
            // the other stage's readers (chunk kc - 1, or the epilogue before) are done
            if (kc + 1 < nchunks) dma_u(kc + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            const float* us = lds + cur * W4_STAGE + (W4_PW * wid) * W4_KC * 32 + lhi * 32 + l31;
            const float* vs = us + W4_OP;
#pragma unroll
            for (int ks = 0; ks < W4_KC / 2; ++ks) {
                float af[W4_PW], bf[W4_PW];
#pragma unroll
                for (int j = 0; j < W4_PW; ++j) {
#ifdef W43_ABL_NOLDSRD
                    af[j] = (float)(kc + j); bf[j] = (float)(ks + j);
#else
                    af[j] = us[(j * W4_KC + 2 * ks) * 32];
                    bf[j] = vs[(j * W4_KC + 2 * ks) * 32];
#endif
                }
                if (kc == 0 && ks == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < W4_PW; ++j) acc[j] = mfma32(af[j], bf[j], zero);
                } else {
#pragma unroll
#ifndef W43_ABL_NOMFMA
                    for (int j = 0; j < W4_PW; ++j) acc[j] = mfma32(af[j], bf[j], acc[j]);
#else
                    for (int j = 0; j < W4_PW; ++j) acc[j][0] += af[j] * bf[j];
#endif
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (xform && kc + 1 < nchunks) {
                write_v(cur ^ 1);  // from the patches requested a whole chunk ago
                if (kc + 2 < nchunks) load_patch(kc + 2);
            }
        }

        // ---- epilogue: M through LDS (stage 1), 16 channels at a time; the next item's chunk 0 is started underneath ----
        // S[(xi * 8 + g) * 64 + h * 32 + tile], channel within the half = (g & 3) + 8 (g >> 2) + 4 h
        float* const S = lds + W4_STAGE;
        const unsigned e_o00 = o00;
        const int e_m0 = m0, e_tb = tb, e_slot = slot, e_ncols = ncols, e_nrows = nrows;
        const bool e_tile_ok = tile_ok;
        const bool has_next = it + 1 < nitems;
        if (has_next) start_item(it + 1);
#ifdef W43_ABL_NOEPI
        {
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < W4_PW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[j][r];
            if (sum == 123.456f) a.dst[0] = sum;
            lds_barrier();
            if (has_next) {
                dma_u(0, 0);
                if (xform) { load_patch(0); write_v(0); if (nchunks > 1) load_patch(1); }
            }
            continue;
        }
#endif
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            lds_barrier();  // ph 0: the K loop's last readers of both stages are done; ph 1: the first half's readers of S
            if (ph == 0 && has_next) {
                dma_u(0, 0);
                if (xform) load_patch(0);
            }
#pragma unroll
            for (int j = 0; j < W4_PW; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = 8 * ph + q;              // accumulator register: channel (r & 3) + 8 (r >> 2) + 4 lhi
                    const int g = (r & 3) + 4 * ((r >> 2) & 1);
                    S[((W4_PW * wid + j) * 8 + g) * 64 + lane] = acc[j][r];
                }
            lds_barrier();
            if (wid < 8) {  // wave-uniform: g = wid, h = lhi
                const int fl = 16 * ph + (wid & 3) + 8 * (wid >> 2) + 4 * lhi;
                const int f = e_m0 + fl;
                const bool f_ok = f < a.M;
                float tt[6][4];  // rows first: tt[i][.] = A^T applied along j
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    float m[6], y[4];
#pragma unroll
                    for (int j = 0; j < 6; ++j) m[j] = S[((6 * i + j) * 8 + wid) * 64 + lane];
                    w43_at(m, y);
#pragma unroll
                    for (int c = 0; c < 4; ++c) tt[i][c] = y[c];
                }
                float o[4][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float m[6] = {tt[0][c], tt[1][c], tt[2][c], tt[3][c], tt[4][c], tt[5][c]};
                    float y[4];
                    w43_at(m, y);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r][c] = y[r];
                }
                if (e_slot >= 0) {  // uniform: a piece of a K-split unit -- raw partial outputs to its scratch slot
                    const unsigned so = (unsigned)(((e_slot * 32 + fl) * 32 + l31) * 64);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        buffer_store_f32x2(buf_f32x2{o[r][0], o[r][1]}, rs_scr, (int)(so + 16u * r), 0, 0);
                        buffer_store_f32x2(buf_f32x2{o[r][2], o[r][3]}, rs_scr, (int)(so + 16u * r + 8u), 0, 0);
                    }
                } else {
                    const unsigned off = ((f_ok && e_tile_ok) ? e_o00 + (unsigned)f * (unsigned)HW * 4u : kOOB);  // per lane: f differs by half-wave
                    if (!RAG) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const unsigned ro = (off >= kOOB) ? kOOB : off + (unsigned)r * row_bytes;
                            buffer_store_f32x2(buf_f32x2{o[r][0], o[r][1]}, rs_dst, (int)ro, 0, 0);
                            buffer_store_f32x2(buf_f32x2{o[r][2], o[r][3]}, rs_dst, (int)(ro >= kOOB ? kOOB : ro + 8u), 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const bool in = off < kOOB && r < e_nrows && c < e_ncols;
                                buffer_store_f32(o[r][c], rs_dst, (int)(in ? off + (unsigned)r * row_bytes + 4u * c : kOOB), 0, 0);
                            }
                    }
                    if (STATS) {  // a half-wave holds channel f for the unit's 32 tiles
                        float sv = 0.f, sq = 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const float v = (RAG && (r >= e_nrows || c >= e_ncols)) ? 0.f : o[r][c];
                                sv += v; sq += v * v;
                            }
                        if (!e_tile_ok) { sv = 0.f; sq = 0.f; }
                        sv = w43_half_sum(sv);
                        sq = w43_half_sum(sq);
                        const unsigned so = (l31 == 31 && f_ok) ? (unsigned)((f * a.tblocks + e_tb) * 8) : kOOB;
                        buffer_store_f32x2(buf_f32x2{sv, sq}, rs_stats, (int)so, 0, 0);
                    }
                }
            }
            // the next item's first patches have had the first half of the epilogue to arrive
            if (ph == 0 && has_next && xform) {
                write_v(0);
                if (nchunks > 1) load_patch(1);
            }
        }
    }
}

// K-split tail of wino43_kernel: the pieces of tail unit u (unit index nunits + u) sit in the scratch slots of the workgroups
// whose chunk ranges met it -- workgroup i's range starts at chunk i * tail_q of the flattened tail; a range that started in
// the previous unit left its SECOND piece here (slot 2 i + 1), every other one its first (slot 2 i). They are added in
// channel order (ascending i), then stored and counted into the batch-norm statistics like the main kernel does for a whole
// unit. One wave per (unit, channel): lane = (tile, upper / lower two rows of its 4 x 4 outputs).
template <bool STATS>
__global__ __launch_bounds__(256) void wino43_tail_fixup_kernel(const Wino43Args a) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int u = blockIdx.x >> 3, fl = (blockIdx.x & 7) * 4 + wid;
    const int NC = a.Jpad / W4_KC;
    const int unit = a.nunits + u;
    const int mb = unit % a.mblocks, tb = unit / a.mblocks;
    const int f = mb * W4_BF + fl;
    const bool f_ok = f < a.M;
    const int tile = lane >> 1, half = lane & 1;
    const unsigned t = (unsigned)tb * W4_BT + (unsigned)tile;
    const bool tile_ok = t < a.T;
    const unsigned per_img = (unsigned)(a.TH * a.TW);
    const unsigned n = tile_ok ? t / per_img : 0u;
    const unsigned rr = tile_ok ? t - n * per_img : 0u;
    const int th = (int)(rr / (unsigned)a.TW), tw = (int)(rr - (unsigned)th * (unsigned)a.TW);
    const int c0 = u * NC, c1 = c0 + NC;
    const int i0 = c0 / a.tail_q, i1 = (c1 - 1) / a.tail_q;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
    for (int i = i0; i <= i1; ++i) {
        const int slot = 2 * i + (i * a.tail_q < c0 ? 1 : 0);
        const float4* v = reinterpret_cast<const float4*>(a.tail_scr + ((size_t)(slot * 32 + fl) * 32 + tile) * 16 + half * 8);
        const float4 x0 = v[0], x1 = v[1];
        r0.x += x0.x; r0.y += x0.y; r0.z += x0.z; r0.w += x0.w;
        r1.x += x1.x; r1.y += x1.y; r1.z += x1.z; r1.w += x1.w;
    }
    const int HW = a.H * a.W;
    const int ncols = a.W - 4 * tw < 4 ? a.W - 4 * tw : 4, row0 = 4 * th + 2 * half;
    float v0[4] = {r0.x, r0.y, r0.z, r0.w}, v1[4] = {r1.x, r1.y, r1.z, r1.w};
    const bool in0 = tile_ok && row0 < a.H, in1 = tile_ok && row0 + 1 < a.H;
    if (f_ok) {
        float* d = a.dst + ((size_t)n * a.M + f) * HW + (size_t)row0 * a.W + 4 * tw;
        if (ncols == 4 && (a.W & 3) == 0) {
            if (in0) *reinterpret_cast<float4*>(d) = r0;
            if (in1) *reinterpret_cast<float4*>(d + a.W) = r1;
        } else {
            for (int c = 0; c < ncols; ++c) {
                if (in0) d[c] = v0[c];
                if (in1) d[a.W + c] = v1[c];
            }
        }
    }
    if (STATS) {
        float sv = 0.f, sq = 0.f;
        for (int c = 0; c < ncols; ++c) {
            if (in0) { sv += v0[c]; sq += v0[c] * v0[c]; }
            if (in1) { sv += v1[c]; sq += v1[c] * v1[c]; }
        }
        sv = wave_sum_dpp(sv);
        sq = wave_sum_dpp(sq);
        if (lane == 63 && f_ok) {
            float* st = a.stats + ((size_t)f * a.tblocks + tb) * 2;
            st[0] = sv; st[1] = sq;
        }
    }
}

__global__ __launch_bounds__(256) void wino43_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int F, int C,
                                                                  int dx_mode, int Jpad, int Mpad) {
    wino43_pack_one(w, u, F, C, dx_mode, Jpad, Mpad, blockIdx.x * 256 + threadIdx.x);
}

// ---- host side ------------------------------------------------------------------------------------------
struct W43Scratch {
    float* p = nullptr;
    size_t cap = 0;
};
static thread_local W43Scratch g_w43_scratch[64];
static float* w43_scratch(size_t floats) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { fprintf(stderr, "[bcnn_hip] device ordinal %d out of range\n", dev); exit(1); }
    W43Scratch& sc = g_w43_scratch[dev];
    if (sc.p == nullptr || sc.cap < floats) {
        if (sc.p) {
            HIP_CHECK(hipStreamSynchronize(current_stream()));
            HIP_CHECK(hipFree(sc.p));
        }
        const size_t cap = floats < (1u << 20) ? (1u << 20) : floats;
        HIP_CHECK(hipMalloc((void**)&sc.p, cap * sizeof(float)));
        sc.cap = cap;
    }
    return sc.p;
}

static thread_local W43Scratch g_w43_tail_scratch[64];  // separate from the U scratch, which the running kernel reads
static float* w43_tail_scratch(size_t floats) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { fprintf(stderr, "[bcnn_hip] device ordinal %d out of range\n", dev); exit(1); }
    W43Scratch& sc = g_w43_tail_scratch[dev];
    if (sc.p == nullptr || sc.cap < floats) {
        if (sc.p) {
            HIP_CHECK(hipStreamSynchronize(current_stream()));
            HIP_CHECK(hipFree(sc.p));
        }
        HIP_CHECK(hipMalloc((void**)&sc.p, floats * sizeof(float)));
        sc.cap = floats;
    }
    return sc.p;
}


static void wino43_run(const float* src, const float* w, float* dst, const ConvShape& s, int dx_mode, ConvStats* stats) {
    Wino43Args a;
    a.src = src; a.dst = dst;
    a.N = s.N; a.J = dx_mode ? s.F : s.C; a.M = dx_mode ? s.C : s.F; a.H = s.H; a.W = s.W;
    a.TH = (s.H + 3) / 4; a.TW = (s.W + 3) / 4;
    a.T = (unsigned)((long long)s.N * a.TH * a.TW);
    a.Jpad = (a.J + W4_KC - 1) / W4_KC * W4_KC;
    a.Mpad = (a.M + W4_BF - 1) / W4_BF * W4_BF;
    a.mblocks = a.Mpad / W4_BF;
    a.tblocks = (int)((a.T + W4_BT - 1) / W4_BT);
    a.nunits = a.mblocks * a.tblocks;
    a.stats = (stats && stats->partials) ? stats->partials : nullptr;
    a.stats_bytes = a.stats ? (unsigned)((size_t)a.M * a.tblocks * 2 * sizeof(float)) : 0u;
    a.src_bytes = (unsigned)((size_t)s.N * a.J * s.HW * 4);
    a.dst_bytes = (unsigned)((size_t)s.N * a.M * s.HW * 4);
    const size_t u_floats = (size_t)W4_NP * a.Jpad * a.Mpad;
    a.upk_bytes = (unsigned)(u_floats * 4);
    float* U = prepack_take(w, PREPACK_WINO, dx_mode, u_floats);  // transformed ahead by bcnn_hip_conv_prepack?
    if (!U) {
        U = w43_scratch(u_floats);
        wino43_pack_weights_kernel<<<ceil_div((long long)a.Jpad * a.Mpad, 256), 256, 0, current_stream()>>>(w, U, s.F, s.C, dx_mode,
                                                                                                          a.Jpad, a.Mpad);
        KERNEL_CHECK();
    }
    a.upk = U;
    const int nblocks = a.nunits;
    const unsigned grid = (unsigned)(nblocks < kCUs ? nblocks : kCUs);  // persistent: one 147 KB workgroup per CU
    // K-split tail: the blocks of a last, partly filled round as chunk ranges dealt out evenly over all CUs -- a CU then does
    // ceil(R * chunks / CUs) chunks in one or two pieces instead of a whole unit. Priced in chunk times (an epilogue ~ 2).
    a.tail_scr = nullptr; a.tail_q = 0; a.tail_units = 0; a.tail_scr_bytes = 0;
    static const int ksplit_on = BCNN_EXP_ENV("BCNN_HIP_NO_WINO_KSPLIT") ? 0 : 1;  // A/B switch of the experiment build
    const int rem = nblocks % (int)grid;
    if (ksplit_on && rem > 0 && nblocks > (int)grid) {
        const int NC = a.Jpad / W4_KC;
        const int q = (int)ceil_div((long long)rem * NC, (long long)grid);
        const double ep = 2.0;
        if (q >= 1 && q <= NC && q + 2 * ep + 2.0 < NC + ep) {
            a.nunits = nblocks - rem;
            a.tail_units = rem;
            a.tail_q = q;
            const size_t scr_floats = (size_t)2 * grid * 32 * 32 * 16;
            a.tail_scr = w43_tail_scratch(scr_floats);
            a.tail_scr_bytes = (unsigned)(scr_floats * sizeof(float));
        }
    }
    const bool rag = (s.H & 3) != 0 || (s.W & 3) != 0;
    trace_kernel(dx_mode ? "wino43_kernel:dx" : "wino43_kernel:fwd");
    if (a.stats && rag) wino43_kernel<true, true><<<grid, 64 * W4_NW, 0, current_stream()>>>(a);
    else if (a.stats) wino43_kernel<true, false><<<grid, 64 * W4_NW, 0, current_stream()>>>(a);
    else if (rag) wino43_kernel<false, true><<<grid, 64 * W4_NW, 0, current_stream()>>>(a);
    else wino43_kernel<false, false><<<grid, 64 * W4_NW, 0, current_stream()>>>(a);
    KERNEL_CHECK();
    if (a.tail_units > 0) {
        trace_kernel("wino43_tail_fixup");
        if (a.stats) wino43_tail_fixup_kernel<true><<<(unsigned)(a.tail_units * 8), 256, 0, current_stream()>>>(a);
        else wino43_tail_fixup_kernel<false><<<(unsigned)(a.tail_units * 8), 256, 0, current_stream()>>>(a);
        KERNEL_CHECK();
    }
    if (stats) stats->splits = a.stats ? a.tblocks : 0;
}

