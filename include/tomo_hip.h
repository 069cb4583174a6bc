/*
 * tomo_hip.h -- C ABI of libtomo_hip.so, the MI355X (gfx950) implementation of the
 * mask stack -> scalar field ("SDF") -> marching-cubes mesh path.
 *
 * The reference (victorramirez952/tomography_3d_reconstructor) has no FFI: its boundary for
 * this path is two Python classes (voxel_processor.py:27-164, surface_extractor.py:28-149)
 * that call NumPy / scipy.ndimage / scikit-image.  This library is what the drop-in classes in
 * tomography_3d_reconstructor_amd/{voxel_processor,surface_extractor}.py bind with ctypes (see
 * INTEGRATION.md); every entry point names the reference line(s) it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer into caller-owned (torch-allocated) memory unless the
 *    parameter name starts with `h_` (host pointer);
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work,
 *    they never synchronise, allocate or free (exceptions are stated);
 *  - return value: 0 = ok, negative = TOMO_E_* (never throws across the ABI);
 *  - volume shape is (nz, ny, nx) = (slices, rows, columns); the padded/field shape is
 *    (Nz, Ny, Nx) = (nz + 2p, ny + 2p, nx + 2p) with p = pad in {0, 1};
 *  - "bits" = bit-packed volume: uint64 words, (nz, ny, wx) with wx = tomo_words_per_row(nx);
 *    bit b of word w of a row is voxel x = 64 w + b; bits at x >= nx are zero.
 */
#ifndef TOMO_HIP_H
#define TOMO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TOMO_OK 0
#define TOMO_E_ARG (-1)       /* bad argument (null pointer, non-positive size, ...) */
#define TOMO_E_LAUNCH (-2)    /* hipGetLastError() after a launch */
#define TOMO_E_SIZE (-3)      /* a count does not fit the index type used */
#define TOMO_E_WORKSPACE (-4) /* workspace too small */

int tomo_abi_version(void);
const char *tomo_error_string(int code);
/* Host-side probe used by the not-gpu tests: evaluates ONE marching-cubes cell with the same
 * code the device runs (compiled for the host).  v = 8 corner values (Lewiner order, float32),
 * tris = up to 36 edge ids (0..12) in LUT order; returns the number of triangles (<0: error). */
int tomo_host_mc_cell(const float *h_v, double iso, int8_t *h_tris, int *h_uses_centre);
double tomo_host_mc_edge_offset(double va, double vb);        /* vertex offset along an edge, same code */
void tomo_host_mc_centre_offset(const double *h_v8, double *h_out3); /* (x,y,z) offset of the centre vertex */

/* Full-content, position-dependent 128-bit checksum of a HOST buffer on `nthreads` host threads (h_out[0..1]).  The
 * device-volume cache of the drop-in classes uses it to make sure a host array the caller could write to still holds
 * what was uploaded (voxel_processor.py:84 / surface_extractor.py:43-46 read the array they are handed). */
int tomo_host_checksum(const void *h_data, int64_t nbytes, int nthreads, uint64_t *h_out);
/* the same with the implementation named (ABI 6): 0 = the fastest the CPU offers (AVX2), 1 = the portable loop; one digest */
int tomo_host_checksum_impl(const void *h_data, int64_t nbytes, int nthreads, int impl, uint64_t *h_out);
/* ... and in two steps (ABI 6), for a buffer that arrives piece by piece (a download in pieces: piece k is digested while piece
 * k + 1 is on the bus): the digests (2 words each) of the tomo_host_checksum_chunk_bytes()-sized chunks of a PART that starts at a
 * chunk boundary, first_chunk = its offset / chunk size; then the fold of all chunk digests of the buffer, in order. */
int64_t tomo_host_checksum_chunk_bytes(void);
int tomo_host_checksum_part(const void *h_part, int64_t nbytes, int64_t first_chunk, int nthreads, int impl, uint64_t *h_dig);
int tomo_host_checksum_fold(const uint64_t *h_dig, int64_t nchunks, int64_t nbytes, uint64_t *h_out);
/* Page a freshly allocated HOST buffer in on `nthreads` threads (one byte per 4 KiB page is written; content unspecified):
 * the 1 B/voxel arrays the drop-in classes hand back (voxel_processor.py:46, :84 create them with np.stack / .copy()) cost
 * ~65 ms per GiB of page faults when a single thread -- or the DMA engine's pinning pass -- touches them first. */
int tomo_host_touch(void *h_data, int64_t nbytes, int nthreads);
/* np.stack (voxel_processor.py:46) on `nthreads` threads: h_dst[i * bytes_each ..] = the bytes_each bytes at h_src[i]. */
int tomo_host_gather(const void *const *h_src, int64_t n, int64_t bytes_each, void *h_dst, int nthreads);
/* SHA-256 with a caller-held, relocatable 112-byte state (ABI 6): the digest of a Z-slab job's WHOLE vertex / face list -- the
 * bytes one np.unique-numbered mesh holds (surface_extractor.py:115-126) -- is formed rank after rank, the state travels
 * between the rank processes instead of the lists.  impl: 0 = fastest available (x86 SHA extensions), 1 = portable code.
 * digest: of everything hashed so far; the state is not changed. */
int tomo_host_sha256_init(void *h_state112);
int tomo_host_sha256_update(void *h_state112, const void *h_data, int64_t nbytes, int impl);
int tomo_host_sha256_digest(const void *h_state112, uint8_t *h_digest32);

/* ---------------------------------------------------------------- geometry helpers (host, pure) */
int64_t tomo_words_per_row(int nx);                      /* ceil(nx / 64) */
/* Extended ("halo") bit volume the field kernel reads: reflect/zero borders materialised. */
int64_t tomo_ext_words_per_row(int nx, int pad);
int64_t tomo_ext_rows(int ny, int pad);                  /* ny + 2 pad + 4 */
int64_t tomo_ext_slices(int nz, int pad);                /* nz + 2 pad + 4 */
/* Field buffer: float32 (Nz, Ny, pitch); padded column X lives at column tomo_field_xorg + X. */
int64_t tomo_field_pitch(int nx, int pad);
int tomo_field_xorg(int pad);                            /* 32 - pad: data column x = 0 sits on a 128-byte line */
int64_t tomo_mc_segments_per_row(int Nx, int xorg);      /* ceil((xorg + Nx + 224) / 256) */

/* ---------------------------------------------------------------- binary stages */
/* np.stack(mask_images) as uint8 0/1 (voxel_processor.py:46) -> bits. */
int tomo_pack_bits(const uint8_t *mask, uint64_t *bits, int nz, int ny, int nx, void *stream);
int tomo_unpack_bits(const uint64_t *bits, uint8_t *mask, int nz, int ny, int nx, void *stream);
/* np.sum(voxel_data) (voxel_processor.py:51): *count (device uint64) += popcount; caller zeroes it. */
int tomo_popcount(const uint64_t *bits, int nz, int ny, int nx, unsigned long long *count, void *stream);
/* ndimage.binary_fill_holes on slice `z` when that slice is non-empty (voxel_processor.py:60-62,66-68).
 * scratch: ny * wx + 8 words.  Iterates on the device until the flood is stable. */
int tomo_fill_holes_slice(uint64_t *bits, int nz, int ny, int nx, int z, uint64_t *scratch, void *stream);
/* The same for slices 0 and nz - 1 in one launch (the two floods are independent); same scratch. */
int tomo_fill_holes_ends(uint64_t *bits, int nz, int ny, int nx, uint64_t *scratch, void *stream);
/* np.stack + _close_volume_ends in one pass over the mask (voxel_processor.py:46, :56-77): the end slices are packed and
 * filled first, then ONE streaming kernel packs every other slice and applies the recurrence, which is the local stencil
 * c'[z] = c[z] | (c[z-1] & c[z+1]).  Needs nz >= 3, nx % 16 == 0 and a 16-byte aligned mask (TOMO_E_ARG otherwise: use
 * tomo_pack_bits + tomo_fill_holes_ends + tomo_close_ends_scan).  scratch: ny * wx + 8 words. */
int tomo_pack_close_ends(const uint8_t *mask, uint64_t *bits, int nz, int ny, int nx, uint64_t *scratch, void *stream);
/* The fused pass for ONE Z-slab of a sharded stack: lo_fixed / hi_fixed say that the slab's first / last slice is a GLOBAL
 * end slice, already packed and filled in `bits`; otherwise `below` / `above` (bit-packed (ny, wx) slices with the ORIGINAL
 * content of the neighbour rank's adjacent slice) close the stencil and every slice of the slab is computed.  nz >= 2. */
int tomo_pack_close_slab(const uint8_t *mask, uint64_t *bits, int nz, int ny, int nx, const uint64_t *below,
                         const uint64_t *above, int lo_fixed, int hi_fixed, void *stream);
/* tomo_pack_close_slab for the output slices [z_from, z_to) only (the middle of a slab depends on the mask alone: a Z-slab
 * rank enqueues it before it talks to its neighbours, the end ranges afterwards; `below` / `above` are needed only if the
 * range reaches an end that is not fixed).  Any split of [0, nz) into ranges gives the same bits. */
int tomo_pack_close_range(const uint8_t *mask, uint64_t *bits, int nz, int ny, int nx, int z_from, int z_to,
                          const uint64_t *below, const uint64_t *above, int lo_fixed, int hi_fixed, void *stream);
/* The stencil on n bit-packed slices whose neighbours are given separately: out[i] = mid[i] | (prev & next), prev = i ?
 * mid[i-1] : before, next = i < n-1 ? mid[i+1] : after ((ny, wx) words per slice; out must not overlap mid).  A Z-slab
 * rank closes the ORIGINAL halo slices of its neighbours with it. */
/* Launch-count savers of the Z-slab front (slab.py; scale-out of voxel_processor.py:46, :56-77, no counterpart in the
 * single-process reference): tomo_pack_bits_pair = two tomo_pack_bits in one launch (the original edge slices for the two
 * neighbours; needs nx % 16 == 0 and 16-byte aligned masks); tomo_slab_edges = tomo_pack_close_range(0, edge) +
 * tomo_pack_close_range(nz - edge, nz) + the two tomo_close_stencil calls on the neighbours' halo slices (lo_n / hi_n = 0:
 * no such neighbour) in ONE launch. */
int tomo_pack_bits_pair(const uint8_t *maskA, uint64_t *bitsA, int nzA, const uint8_t *maskB, uint64_t *bitsB, int nzB,
                        int ny, int nx, void *stream);
int tomo_slab_edges(const uint8_t *mask, uint64_t *bits, int nz, int ny, int nx, int edge, const uint64_t *below,
                    const uint64_t *above, int lo_fixed, int hi_fixed, const uint64_t *lo_before, const uint64_t *lo_mid,
                    const uint64_t *lo_after, int lo_n, uint64_t *lo_out, const uint64_t *hi_before, const uint64_t *hi_mid,
                    const uint64_t *hi_after, int hi_n, uint64_t *hi_out, void *stream);
int tomo_close_stencil(const uint64_t *before, const uint64_t *mid, const uint64_t *after, int n, int ny, int nx,
                       uint64_t *out, void *stream);
/* The z recurrence of _close_volume_ends (voxel_processor.py:72-75), in place.
 * workspace: tomo_close_ends_workspace_words() uint64 words. */
int64_t tomo_close_ends_workspace_words(int nz, int ny, int nx);
int tomo_close_ends_scan(uint64_t *bits, int nz, int ny, int nx, uint64_t *workspace, void *stream);
/* The same chain reduced to a pair of bit planes gp_out = [G | P] (2 * ny * wx words) over slices 1 .. nz-2:
 * c'[nz-2] = G | (P & c'[0]).  What a Z-slab rank publishes in the multi-GPU path (same workspace size). */
int tomo_close_ends_gp(const uint64_t *bits, int nz, int ny, int nx, uint64_t *workspace, uint64_t *gp_out, void *stream);
/* One 6-neighbour pass (skimage binary_erosion/binary_dilation, voxel_processor.py:88,91):
 * op 0 = erosion with border_value 1, op 1 = dilation with border value 0.  in != out. */
int tomo_morph_pass(const uint64_t *in, uint64_t *out, int nz, int ny, int nx, int op, void *stream);
/* nops (2, 4, 6 or 8) such passes fused into one kernel: bit j of `ops` is the op of pass j (0 erosion, 1 dilation).
 * smooth_voxel_data(iterations=3, create_manifold=True) is E D | D E | D E | D E = ops 0b01010110. */
int tomo_morph_fused(const uint64_t *in, uint64_t *out, int nz, int ny, int nx, uint32_t ops, int nops, void *stream);

/* ---------------------------------------------------------------- callers either side of the path */
/* volume_calculator.py:23-35: counts[z] (device uint64[nz]) = np.sum(voxel_data[z]); the call zeroes counts first. */
int tomo_slice_popcounts(const uint64_t *bits, int nz, int ny, int nx, unsigned long long *counts, void *stream);
/* volume_calculator.py:40,62 (np.where(voxel_data) + min/max): box (device int32[6]) = {zmin, zmax, ymin, ymax, xmin,
 * xmax} of the set voxels; an empty volume gives {INT32_MAX, -1, INT32_MAX, -1, INT32_MAX, -1}. */
int tomo_bbox(const uint64_t *bits, int nz, int ny, int nx, int32_t *box, void *stream);
/* image_loader.py:108 (`img >= threshold`) fused with the packing: grey = uint8 (nz, ny, nx) on the device. */
int tomo_pack_threshold(const uint8_t *grey, uint64_t *bits, int nz, int ny, int nx, int threshold, void *stream);
/* obj_exporter.py:17-38, byte for byte ("v %.6f %.6f %.6f" per vertex, "f a+1 b+1 c+1" per face), HOST arrays:
 * vertices nv x 3 float32 (vertex_is_double = 0) or float64 (1), faces nf x 3 int64, 0-based.  Formats in parallel
 * on `nthreads` host threads.  Returns 0, TOMO_E_ARG, or -errno when the file cannot be written. */
int tomo_obj_write(const char *path, const void *vertices, int vertex_is_double, int64_t nv, const int64_t *faces,
                   int64_t nf, int nthreads);
/* The same file written by several processes (a Z-slab job; BASELINE configs[3] "seam-free OBJ export"): every rank
 * formats ITS run of the vertex list (kind 0: float32 rows, 1: float64 rows) or of the face list (kind 2: int64 rows
 * holding GLOBAL 0-based vertex indices) in host memory, learns the size, and -- once the sizes of all ranks are known
 * -- writes the bytes at its offset of the shared file, which must exist.  Same formatter as tomo_obj_write: the file
 * equals the one written from the gathered mesh.  A block lives until tomo_obj_block_free. */
int tomo_obj_block_format(int kind, const void *h_rows, int64_t n, int nthreads, void **h_block, int64_t *h_nbytes);
int tomo_obj_block_pwrite(const char *path, int64_t offset, const void *h_block);
void tomo_obj_block_free(void *h_block);

/* ---------------------------------------------------------------- scalar field ("SDF") */
/* bits -> extended bits (reflect of the padded array + zero pad ring), see tomo_ext_*. */
int tomo_extend_bits(const uint64_t *bits, uint64_t *ext, int nz, int ny, int nx, int pad, void *stream);
/* surface_extractor.py:43-53 + the float32 cast of skimage's wrapper: pad, astype(float64),
 * gaussian_filter(sigma=0.5) (three 5-tap float64 correlate1d passes, axis 0,1,2, mode reflect),
 * cast to float32.  gaussian = 0 writes the raw 0/1 field (manifold=False). */
int tomo_field_fill(const uint64_t *ext, float *field, int nz, int ny, int nx, int pad, int gaussian,
                    unsigned long long *signs, uint8_t *gcls, void *stream);   /* `field` must be 128-byte aligned */
/* The Gaussian field straight from the plain bit volume (no tomo_extend_bits): the block forms the extended words
 * while it stages its input.  Same outputs as tomo_field_fill(gaussian = 1) on tomo_extend_bits(bits). */
int tomo_field_fill_bits(const uint64_t *bits, float *field, int nz, int ny, int nx, int pad, unsigned long long *signs,
                         uint8_t *gcls, void *stream);
/* The same, minus what nothing downstream can read: the float field is an intermediate of surface_extractor.py:43-55 that
 * marching cubes reads only at the corners of active cells, so constant tiles whose input is uniform within 3 voxels stay
 * unwritten (their floats are undefined afterwards).  Sign records / classes as above (both required here).  span_ws:
 * 4-byte aligned device scratch of tomo_field_span_bytes(nz, ny, nx, pad) bytes. */
int64_t tomo_field_span_bytes(int nz, int ny, int nx, int pad);
int tomo_field_fill_bits_sparse(const uint64_t *bits, float *field, int nz, int ny, int nx, int pad, unsigned long long *signs,
                                uint8_t *gcls, uint8_t *span_ws, void *stream);
/* Sign records (input of marching-cubes pass 1): uint64 [Nz][S][NyP][4], S = tomo_mc_segments_per_row(Nx, xorg),
 * NyP = tomo_sign_rows(Ny) (Ny rounded up to 16 so that 16-row groups of records are 512-byte aligned);
 * bit L of word k of record (Z, s, Y) = [field(Z, Y, column 256 s - 224 + 4 L + k) > iso].  tomo_field_fill writes
 * them as a by-product (iso 0.5) when `signs` is not NULL and gaussian = 1 (no zeroing needed; bits of columns outside
 * the padded row are unspecified and ignored by tomo_mc_classify).  Records come in GROUPS of 16 rows; gcls
 * (uint8 [Nz][NyP / 16][S], always passed together with signs) holds the class of every group: 0 / 1 = all bits 0 / 1
 * and the 16 records are NOT stored (70 % of the groups of an ellipsoid volume), 2 = the records are stored.
 * tomo_field_signs derives records from any float field for slices [z_begin, z_end) (all groups of class 2). */
int64_t tomo_sign_rows(int Ny);
int tomo_field_signs_fused(int nx);                      /* 1: tomo_field_fill writes the records itself (always, ABI 2) */
/* Size (uint64 words) of the buffer passed as `signs` to tomo_field_fill. */
int64_t tomo_sign_buffer_words(int Nz, int Ny, int Nx, int xorg);
int tomo_field_signs(const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso, int z_begin,
                     int z_end, unsigned long long *signs, uint8_t *gcls, void *stream);

/* ---------------------------------------------------------------- marching cubes (Lewiner MC33) */
/* skimage.measure.marching_cubes(volume, level) (surface_extractor.py:55) as five device passes.
 * A SEGMENT = 256 consecutive float columns of a field row (voxel X is in segment (X + xorg + 224) / 256);
 * nseg = Nz * Ny * tomo_mc_segments_per_row(Nx, xorg).
 * A voxel is ACTIVE when its 8 cube corners (neighbours clamped at the volume border) are not all on one side
 * of `iso` (a corner equal to iso counts as below, like the reference).
 * vertex/voxel key = (row << 22) | (X << 2) | slot, row = Z*Ny + Y; slot 0/1/2 = x/y/z edge owned by the
 * voxel, 3 = cell-centre vertex (voxel keys have slot 0).
 *
 * 1. classify: from the sign records, seg_cnt[seg] = number of active voxels of EVERY segment (uint32[nseg],
 *    seg = (Z*Ny + Y)*S + s) and, for every NON-EMPTY segment, seg_act[4*seg + k] = 64-bit mask whose bit L says
 *    voxel 4L+k of the segment (X = 256 s - 224 + 4L + k - xorg) is active (seg_act: uint64[4*nseg]; records of
 *    empty segments stay unwritten and are never read).  No float is read. */
int tomo_mc_classify(const unsigned long long *signs, const uint8_t *gcls, int Nz, int Ny, int Nx, int xorg,
                     unsigned long long *seg_act, uint32_t *seg_cnt, void *stream);
/* Segment-level scan: seg_aoff uint32[nseg + 1] = exclusive scan of seg_cnt, totals (device uint64[4]) =
 * {active voxels, 0, 0, 0}.  workspace: tomo_mc_scan_workspace_bytes(nseg) bytes. */
int tomo_mc_scan_segments(const uint32_t *seg_cnt, int64_t nseg, uint32_t *seg_aoff, unsigned long long *totals,
                          void *workspace, int64_t workspace_bytes, void *stream);
/* Exclusive scan of packed counts (low 16 bits -> off_a, high 16 bits -> off_b; both uint32[n + 1]; with off_b = NULL
 * the counts are plain numbers and off_a their exclusive scan), totals (device uint64[4]) = {sum low, sum high, 0, 0}.
 * nz_ids must be NULL (kept for the call shape).  workspace: tomo_mc_scan_workspace_bytes(n) bytes. */
int64_t tomo_mc_scan_workspace_bytes(int64_t n);
int tomo_mc_scan(const uint32_t *counts, int64_t n, uint32_t *off_a, uint32_t *off_b, uint32_t *nz_ids,
                 unsigned long long *totals, void *workspace, int64_t workspace_bytes, void *stream);
/* 2. list: vox_key[na] = keys of the active voxels, ascending (cell scan order), from seg_act. */
int tomo_mc_list(int Nz, int Ny, int Nx, int xorg, const uint32_t *seg_aoff, const unsigned long long *seg_act,
                 unsigned long long *vox_key, void *stream);
/* 3. eval: one MC33 evaluation per active voxel: vox_counts[na] = ntri << 16 | nvert,
 * vox_flags[na] = bit0/1/2 edge vertices, bit3 centre vertex. */
int tomo_mc_eval(const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso,
                 const unsigned long long *vox_key, int64_t na, uint32_t *vox_counts, uint8_t *vox_flags, void *stream);
/* Capped variants: the caller launches them BEFORE it has read the segment scan's total (one host round trip less per
 * pass).  `cap` entries of buffer; list: segments that would not fit are skipped; eval: entries at and beyond *na_dev
 * (device memory) get count 0, so tomo_mc_scan over all `cap` counts yields the same totals.  The caller checks
 * na <= cap afterwards and falls back to the exact calls otherwise. */
int tomo_mc_list_capped(int Nz, int Ny, int Nx, int xorg, const uint32_t *seg_aoff, const unsigned long long *seg_act,
                        unsigned long long *vox_key, int64_t cap, void *stream);
int tomo_mc_eval_capped(const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso,
                        const unsigned long long *vox_key, int64_t cap, const unsigned long long *na_dev,
                        uint32_t *vox_counts, uint8_t *vox_flags, void *stream);
/* 4. emit: vertices (key + raw MC position, (z,y,x) float32 as skimage returns them; keys ascending) and
 * triangles as provisional vertex indices (int32), in the reference's order and with the per-triangle
 * reversal of skimage/measure/_marching_cubes_lewiner.py:338.  totals[3] counts corners whose vertex was
 * not found (must stay 0).  z_offset (0 on one GPU) is added to the slice index of every vertex position before
 * rounding: a Z-slab rank emits positions in global padded coordinates. */
int tomo_mc_emit(const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso,
                 const unsigned long long *vox_key, int64_t na, const unsigned long long *seg_act,
                 const uint32_t *seg_aoff, const uint32_t *vox_voff, const uint32_t *vox_foff, const uint8_t *vox_flags,
                 int z_offset, unsigned long long *vkey, float *vpos, int32_t *faces, unsigned long long *totals,
                 void *stream);   /* seg_act / seg_aoff: from classify / scan_segments (vertex lookup by ballot rank) */

/* 5. (manifold=False only) skimage's first-touch vertex numbering, which the reference returns unchanged when it
 * skips np.unique (surface_extractor.py:67-68): mode 0 writes created[na] = vertices each cell creates in the
 * serial scan; after an exclusive scan (tomo_mc_scan) mode 1 writes ft_rank[provisional vertex] = its number. */
int tomo_mc_first_touch(const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso,
                        const unsigned long long *vox_key, int64_t na, const unsigned long long *seg_act,
                        const uint32_t *seg_aoff, const uint32_t *vox_voff, const uint8_t *vox_flags, int mode,
                        uint32_t *created, const uint32_t *base, int32_t *ft_rank, unsigned long long *totals, void *stream);

/* ---------------------------------------------------------------- "mc3": marching cubes + finalisation + unique in one chain
 * The production path for manifold=True (surface_extractor.py:55-72): same inputs as above (sign records -> tomo_mc_classify
 * -> seg_cnt / seg_act), then
 *   tomo_mc3_list      block sums of seg_cnt, one single-workgroup scan, list of active voxels; tot[0] = list length
 *   tomo_mc3_eval      one float64 evaluation per active voxel: MC33 tiling reference, vertex flags, vertex coordinates
 *                      along the owned edges (float32), counts reduced per block of 256 entries
 *   tomo_mc3_scan      single workgroup: block prefixes, totals (tot[1] vertices, tot[2] triangles), per-slice tables and
 *                      the offsets of the sort's segments (slice_tab: tomo_mc3_slice_table_words(Nz, Ny) uint32)
 *   tomo_mc3_vertices  FINAL vertex rows (-1 shift, slice-depth map, y / x scale: surface_extractor.py:57-65, :82-113) as
 *                      16-byte records {z', y', x', id}, partitioned per slice into in-plane / between-plane buckets,
 *                      with their 32-bit sort keys; vertex id = 4 * (list position of the owner voxel) + slot
 *   tomo_mc3_sort_rank segmented sort inside the buckets + gather: uniq rows in np.unique's order, table[id] = index;
 *                      tot[4] counts the places where the rows do not ascend strictly (0 <=> the result is exact)
 *   tomo_mc3_faces     final int64 triangles (reference order and winding) through table; tot[5] degenerate triangles
 *                      (the caller drops them), tot[6] corners without vertex (must stay 0)
 * Every kernel takes the list length / totals from `tot` (device uint64[8]): the chain can be enqueued into buffers sized
 * from a hint (cap list entries, cap_v vertices, cap_f triangles) before any count is known to the host; what does not fit
 * is flagged in tot[3] (1 list, 2 vertices, 4 triangles) and nothing is written past a buffer.
 * Buffers: seg_blk uint32[ceil(nseg / 256)], seg_aoff uint32[nseg + 1], vox_key uint64[cap], vox_loc uint32[cap],
 * vox_til int32[cap], vox_flags uint8[cap], vox_used uint16[cap] (cube edges with a vertex), vox_f3 / vox_c3 float[3 cap], blk3 uint32[3 ceil(cap / 256)],
 * vrec float[4 cap_v], keys / idx uint32[cap_v], uniq float[3 cap_v], table int32[4 cap], faces int64[3 cap_f]. */
int tomo_mc3_list(int Nz, int Ny, int Nx, int xorg, const uint32_t *seg_cnt, const unsigned long long *seg_act,
                  uint32_t *seg_blk, uint32_t *seg_aoff, unsigned long long *vox_key, int64_t cap, unsigned long long *tot,
                  void *stream);
int tomo_mc3_eval(const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso,
                  const unsigned long long *vox_key, int64_t cap, const unsigned long long *tot, int z_offset,
                  uint32_t *vox_loc, int32_t *vox_til, uint8_t *vox_flags, uint16_t *vox_used, float *vox_f3, float *vox_c3,
                  uint32_t *blk3, void *stream);
int64_t tomo_mc3_slice_table_words(int Nz, int Ny);
int64_t tomo_mc3_sort_segments(int Nz, int Ny);      /* sort segments: per slice its plane (cut into bands of 512 owner rows when Ny > 1280) + the between-plane bucket */
int tomo_mc3_scan(int Nz, int Ny, int Nx, int xorg, const uint32_t *seg_aoff, const uint32_t *vox_loc, int64_t cap,
                  uint32_t *blk3, uint32_t *slice_tab, unsigned long long *tot, int64_t cap_v, int64_t cap_f, void *stream);
int tomo_mc3_vertices(int Nz, int Ny, int Nx, int xorg, const unsigned long long *vox_key, int64_t cap,
                      const unsigned long long *tot, const uint32_t *vox_loc, const uint8_t *vox_flags, const float *vox_f3,
                      const float *vox_c3, const uint32_t *blk3, const uint32_t *slice_tab, int z_offset, int shift,
                      const double *cum, int64_t ncum, const double *adj, int64_t nadj, float mm_y, float mm_x, float *vrec,
                      uint32_t *keys, uint32_t *idx, void *stream);
int64_t tomo_mc3_sort_workspace_bytes(int64_t cap_v, int64_t nseg);    /* nseg = tomo_mc3_sort_segments(Nz, Ny) */
int tomo_mc3_sort_rank(const float *vrec, uint32_t *keys, uint32_t *idx, int64_t cap_v, int Nz, int Ny, const uint32_t *slice_tab,
                       unsigned long long *tot, float *uniq, int32_t *table, void *workspace, int64_t workspace_bytes,
                       void *stream);
/* The same, and tot[7] = number of rows with z' == z_top (the plane a Z-slab rank shares with the rank above; NaN: none). */
int tomo_mc3_sort_rank_top(const float *vrec, uint32_t *keys, uint32_t *idx, int64_t cap_v, int Nz, int Ny, const uint32_t *slice_tab,
                           unsigned long long *tot, float *uniq, int32_t *table, void *workspace, int64_t workspace_bytes,
                           float z_top, void *stream);
/* ABI 6, the default since round 4: the same stage -- np.unique(axis=0)'s order inside the buckets, rows, table, the count of
 * places that do not ascend strictly (tot[4]), the rows on z' == z_top (tot[7]) -- in ONE hand-written kernel (a workgroup per
 * segment: wave bitonic + merge rounds in LDS, rows gathered in sorted order), no library primitive, no workspace; `idx` of
 * tomo_mc3_vertices may be NULL for it.  A segment longer than 4 096 entries (1 024 for the clamped run of a padded stack's
 * first slices) sets bit 8 of tot[3]: repeat the stage with tomo_mc3_sort_rank_top (after zeroing tot[3], tot[4], tot[7]). */
int tomo_mc3_sort_rank_fused(const float *vrec, const uint32_t *keys, int64_t cap_v, int Nz, int Ny, uint32_t *slice_tab,
                             unsigned long long *tot, float *uniq, int32_t *table, float z_top, void *stream);
int tomo_mc3_faces(int Nz, int Ny, int Nx, int xorg, const unsigned long long *vox_key, int64_t cap, unsigned long long *tot,
                   const unsigned long long *seg_act, const uint32_t *seg_aoff, const uint32_t *vox_loc, const int32_t *vox_til,
                   const uint16_t *vox_used, const uint32_t *blk3, const int32_t *table, int64_t *faces, int64_t cap_f,
                   void *stream);

/* ---------------------------------------------------------------- mesh finalisation */
/* surface_extractor.py:57-65 + :82-113 on (V,3) float32 rows in place: -1 shift (if shift),
 * variable slice depth map of z (cum/adj float64 tables as the reference builds them; nadj = 0
 * skips it), y *= mm_y, x *= mm_x (float32). */
int tomo_vertex_finalize(float *vpos, int64_t nv, int shift, const double *cum, int64_t ncum,
                         const double *adj, int64_t nadj, float mm_y, float mm_x, void *stream);
/* surface_extractor.py:115-126 (np.unique(axis=0, return_inverse) + drop faces with < 3 distinct
 * indices), on the device:
 *   tomo_mesh_unique: lexicographic (z,y,x) sort of the vertex rows, dedupe -> uniq (U,3),
 *                     rank[V] = final index of every provisional vertex; totals[0] = U;
 *   tomo_mesh_faces:  faces32 -> rank -> degenerate triangles dropped (order kept) -> int64;
 *                     totals[1] = number of kept faces.
 * Workspace sizes from the *_bytes helpers. */
int64_t tomo_mesh_unique_workspace_bytes(int64_t nv);
int tomo_mesh_unique(const float *vpos, int64_t nv, float *uniq, int32_t *rank, unsigned long long *totals,
                     void *workspace, int64_t workspace_bytes, void *stream);
/* The same through the one-sort path, for rows in marching-cubes order together with the vertex keys tomo_mc_emit wrote
 * (vkey) and the field's Ny / Nz: the order is that of ONE stable sort on a 48-bit key (slice bucket, then y in a plane or
 * z between planes), carried out as a stable two-way partition inside every slab plus a segmented 32-bit sort inside
 * the 2 Nz buckets.  Exact if and only if totals[2] (which the caller zeroes) is still 0 afterwards -- it counts the places
 * where the result descends in (z, y, x) (float32 rounding coincidences, zero slice depths); if it is not 0, call
 * tomo_mesh_unique instead.  Same workspace. */
int tomo_mesh_unique_presorted(const float *vpos, const unsigned long long *vkey, int64_t nv, int Ny, int Nz, float *uniq,
                               int32_t *rank, unsigned long long *totals, void *workspace, int64_t workspace_bytes,
                               void *stream);
int64_t tomo_mesh_faces_workspace_bytes(int64_t nf);
/* out[i] = index of query row i in the sorted duplicate-free row list uniq (nu x 3), -1 (and *missing += 1, a device
 * counter the caller zeroes) if it is not there.  Used by the Z-slab job for the shared-plane vertices. */
int tomo_mesh_lookup(const float *uniq, int64_t nu, const float *query, int64_t nq, int32_t *out,
                     unsigned long long *missing, void *stream);
/* Z-slab numbering without host round trips (scale-out of the path; no counterpart in the single-process
 * reference): all counts come from `tot` of the rank's mc3 chain (tot[1] rows, tot[7] of them on the shared top plane).
 *   tomo_slab_top_rows  msg float32[(cap + 1) * 3]: row 0 = {n_top as uint32 bits, 0, 0}, then the top-plane rows, zero padded
 *   tomo_slab_lookup    out[i] = index of row i of a received message in this rank's uniq, -1 past its count or when the
 *                       row is not there (then *missing += 1; zero before the first call -- tomo_slab_summary reads it and
 *                       clears it again)
 *   tomo_slab_summary   out int64[8] = kept rows | missing | flags (1 chain overflow, 2 rows not strictly ascending,
 *                       4 n_top > cap_top, 8 caller_flags != 0) | rows | top rows | rows announced from below | list length |
 *                       triangles -- what the ranks all-gather
 *   tomo_mc3_faces_slab tomo_mc3_faces with GLOBAL indices: row r of this rank's list (what `table` names) leaves as r + this
 *                       rank's offset (the sum of the lower ranks' kept rows in `gathered`, int64[world][8]) while r < kept
 *                       rows, and as ids_next[r - kept] (what the upper rank's tomo_slab_lookup found) + the upper rank's
 *                       offset for the rows on the shared top plane */
int tomo_slab_top_rows(const float *uniq, const unsigned long long *tot, int64_t cap_v, int64_t cap, float *msg, void *stream);
int tomo_slab_lookup(const float *uniq, const unsigned long long *tot, int64_t cap_v, const float *msg, int64_t cap,
                     int32_t *out, unsigned long long *missing, void *stream);
int tomo_slab_summary(const unsigned long long *tot, int64_t cap_v, const float *msg_in, unsigned long long *missing,
                      int64_t cap_top, int64_t caller_flags, int64_t *out, void *stream);
/* tomo_slab_lookup + tomo_slab_summary in ONE launch (the workgroup that finishes last writes the summary).  scratch: uint64[2],
 * zeroed once by the caller, left zero by every call.  msg == NULL (the lowest rank): the summary alone. */
int tomo_slab_lookup_summary(const float *uniq, const unsigned long long *tot, int64_t cap_v, const float *msg, int64_t cap,
                             int32_t *out, int64_t cap_top, int64_t caller_flags, unsigned long long *scratch, int64_t *summary,
                             void *stream);
int tomo_mc3_faces_slab(int Nz, int Ny, int Nx, int xorg, const unsigned long long *vox_key, int64_t cap, unsigned long long *tot,
                        const unsigned long long *seg_act, const uint32_t *seg_aoff, const uint32_t *vox_loc, const int32_t *vox_til,
                        const uint16_t *vox_used, const uint32_t *blk3, const int32_t *table, int64_t *faces, int64_t cap_f,
                        const int64_t *gathered, int rank, int world, const int32_t *ids_next, int64_t cap_top, int64_t cap_v,
                        void *stream);
int tomo_mesh_faces(const int32_t *faces32, int64_t nf, const int32_t *rank, int64_t *faces_out,
                    unsigned long long *totals, void *workspace, int64_t workspace_bytes, void *stream);
/* Speculative one-pass variant: faces_out[f] = rank[faces32[f]] for every face (int64), totals[1] = nf, and totals[3]
 * (zeroed by the caller) += number of degenerate faces.  The result is final if and only if totals[3] stays 0; otherwise
 * call tomo_mesh_faces. */
int tomo_mesh_faces_direct(const int32_t *faces32, int64_t nf, const int32_t *rank, int64_t *faces_out,
                           unsigned long long *totals, void *stream);
/* surface_extractor.py:128-149: out[0] = sum over faces of dot(v0, cross(v1,v2))/6 (float64
 * accumulation of float32 terms), out[1] = sum of 0.5*|cross(v1-v0, v2-v0)|.  out is zeroed by the
 * caller; tree reduction => parity with the reference's sequential sums is to 1e-6 rel, not bitwise. */
int tomo_mesh_volume_area(const float *verts, const int64_t *faces, int64_t nf, double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TOMO_HIP_H */
