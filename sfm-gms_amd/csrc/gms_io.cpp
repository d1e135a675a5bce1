// gms_io.cpp -- the ingest format (SURVEY.md 8f "f2"): the reference keeps everything in process (std::vector<cv::KeyPoint>,
// cv::Mat descriptors, std::vector<cv::DMatch>; FeatureMatchUtil.cpp:9-12,58-68) and has no on-disk form, so a sequence's
// detector / matcher output cannot reach another process or machine. One little-endian file holds what the batch API takes:
//
//   char     magic[8]  "GMSFRM01"
//   uint32   n_frames, desc_kind (GMS_DESC_NONE / GMS_DESC_HAMMING256 / GMS_DESC_L2_F32X128)
//   uint64   total_kp, n_pairs, total_matches
//   int32    wh[2 * n_frames]             image sizes (cv::Size)
//   int64    frame_off[n_frames + 1]      keypoint offsets
//   28 B   x total_kp                     cv::KeyPoint records, verbatim
//   row    x total_kp                     descriptors (32 B or 512 B rows, one per keypoint; absent for GMS_DESC_NONE)
//   24 B   x n_pairs                      gms_pair
//   16 B   x total_matches                cv::DMatch records, verbatim
//
// A caller of the reference dumps its vectors with gms_dataset_write (the PODs are layout-identical: no conversion), the harness
// reads them back with gms_dataset_read or the numpy mirror in sfm-gms_amd/io.py. Plain host code: no GPU involved.
#define _FILE_OFFSET_BITS 64
#include <cstdio>
#include <sys/types.h>
#include <cstdlib>
#include <cstring>

#include "gms.h"

namespace {
const char kMagic[8] = {'G', 'M', 'S', 'F', 'R', 'M', '0', '1'};
struct Header {
    char magic[8];
    uint32_t n_frames, desc_kind;
    uint64_t total_kp, n_pairs, total_matches;
};
static_assert(sizeof(Header) == 40, "file header layout");

size_t desc_row_bytes(uint32_t kind)
{
    return kind == GMS_DESC_HAMMING256 + 1u ? 32 : kind == GMS_DESC_L2_F32X128 + 1u ? 512 : 0;
}
bool put(FILE* f, const void* p, size_t n) { return n == 0 || std::fwrite(p, 1, n, f) == n; }
bool get(FILE* f, void* p, size_t n) { return n == 0 || std::fread(p, 1, n, f) == n; }
}  // namespace

extern "C" {

int gms_dataset_write(const char* path, const gms_dataset* d)
{
    if (!path || !d || d->n_frames < 0 || d->n_pairs < 0 || d->total_matches < 0) return GMS_ERR_BAD_ARG;
    if (d->n_frames > 0 && (!d->wh || !d->frame_off)) return GMS_ERR_BAD_ARG;
    const int64_t total_kp = d->n_frames > 0 ? d->frame_off[d->n_frames] : 0;
    if (total_kp < 0 || (total_kp > 0 && !d->keypoints) || (d->n_pairs > 0 && !d->pairs) || (d->total_matches > 0 && !d->matches))
        return GMS_ERR_BAD_ARG;
    if (d->desc_kind != GMS_DESC_NONE && d->desc_kind != GMS_DESC_HAMMING256 && d->desc_kind != GMS_DESC_L2_F32X128) return GMS_ERR_BAD_ARG;
    if (d->desc_kind != GMS_DESC_NONE && total_kp > 0 && !d->descriptors) return GMS_ERR_BAD_ARG;
    FILE* f = std::fopen(path, "wb");
    if (!f) return GMS_ERR_IO;
    Header h;
    std::memcpy(h.magic, kMagic, 8);
    h.n_frames = (uint32_t)d->n_frames;
    h.desc_kind = (uint32_t)(d->desc_kind + 1);  // 0 = none on disk
    h.total_kp = (uint64_t)total_kp;
    h.n_pairs = (uint64_t)d->n_pairs;
    h.total_matches = (uint64_t)d->total_matches;
    const bool ok = put(f, &h, sizeof h) && put(f, d->wh, (size_t)d->n_frames * 8) &&
                    put(f, d->frame_off, d->n_frames > 0 ? (size_t)(d->n_frames + 1) * 8 : 0) &&
                    put(f, d->keypoints, (size_t)total_kp * sizeof(gms_keypoint)) &&
                    put(f, d->descriptors, (size_t)total_kp * desc_row_bytes(h.desc_kind)) &&
                    put(f, d->pairs, (size_t)d->n_pairs * sizeof(gms_pair)) &&
                    put(f, d->matches, (size_t)d->total_matches * sizeof(gms_dmatch));
    const bool closed = std::fclose(f) == 0;
    return ok && closed ? GMS_OK : GMS_ERR_IO;
}

int gms_dataset_read(const char* path, gms_dataset* d)
{
    if (!path || !d) return GMS_ERR_BAD_ARG;
    std::memset(d, 0, sizeof *d);
    FILE* f = std::fopen(path, "rb");
    if (!f) return GMS_ERR_IO;
    Header h;
    if (!get(f, &h, sizeof h) || std::memcmp(h.magic, kMagic, 8) != 0 || h.desc_kind > 2u || h.n_frames > (1u << 30)) {
        std::fclose(f);
        return GMS_ERR_IO;
    }
    const size_t row = desc_row_bytes(h.desc_kind);
    // The counts come from the file: bound each (2^40 elements: far beyond anything that fits a machine, far below where the byte
    // sizes could wrap 64 bits) and require the arrays they describe to be exactly what follows the header, before anything is allocated.
    const uint64_t kMaxCount = (uint64_t)1 << 40;
    if (h.total_kp > kMaxCount || h.n_pairs > kMaxCount || h.total_matches > kMaxCount) {
        std::fclose(f);
        return GMS_ERR_IO;
    }
    const size_t sz_wh = (size_t)h.n_frames * 8, sz_off = h.n_frames ? (size_t)(h.n_frames + 1) * 8 : 0;
    const size_t sz_kp = (size_t)h.total_kp * sizeof(gms_keypoint), sz_desc = (size_t)h.total_kp * row;
    const size_t sz_pairs = (size_t)h.n_pairs * sizeof(gms_pair), sz_m = (size_t)h.total_matches * sizeof(gms_dmatch);
    {
        const uint64_t payload = (uint64_t)sz_wh + sz_off + sz_kp + sz_desc + sz_pairs + sz_m;  // each term < 2^50: no wrap
        long long file_size = -1;
        if (fseeko(f, 0, SEEK_END) == 0) file_size = (long long)ftello(f);
        if (file_size < 0 || (uint64_t)file_size != sizeof(Header) + payload || fseeko(f, (off_t)sizeof(Header), SEEK_SET) != 0) {
            std::fclose(f);
            return GMS_ERR_IO;  // truncated, padded, or a header that does not describe this file
        }
    }
    // one block, every array 16-byte aligned inside it
    auto up = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t total = up(sz_wh) + up(sz_off) + up(sz_kp) + up(sz_desc) + up(sz_pairs) + up(sz_m) + 16;
    char* block = (char*)std::malloc(total);
    if (!block) {
        std::fclose(f);
        return GMS_ERR_IO;
    }
    char* p = block;
    auto take = [&](size_t n) { char* q = p; p += up(n); return q; };
    char* a_wh = take(sz_wh), *a_off = take(sz_off), *a_kp = take(sz_kp), *a_desc = take(sz_desc), *a_pairs = take(sz_pairs), *a_m = take(sz_m);
    bool ok = get(f, a_wh, sz_wh) && get(f, a_off, sz_off) && get(f, a_kp, sz_kp) && get(f, a_desc, sz_desc) && get(f, a_pairs, sz_pairs) &&
              get(f, a_m, sz_m);
    std::fclose(f);
    if (ok && h.n_frames) {  // the offsets must describe the arrays that follow
        const int64_t* off = (const int64_t*)a_off;
        ok = off[0] == 0 && (uint64_t)off[h.n_frames] == h.total_kp;
        for (uint32_t i = 0; ok && i < h.n_frames; ++i) ok = off[i] <= off[i + 1];
    } else if (ok) {
        ok = h.total_kp == 0;  // keypoints without frames
    }
    if (ok) {  // every pair must name frames of the file and a range of its match array
        const gms_pair* pr = (const gms_pair*)a_pairs;
        for (uint64_t i = 0; ok && i < h.n_pairs; ++i)
            ok = pr[i].frame_a >= 0 && (uint32_t)pr[i].frame_a < h.n_frames && pr[i].frame_b >= 0 && (uint32_t)pr[i].frame_b < h.n_frames &&
                 pr[i].m >= 0 && pr[i].match_off >= 0 && (uint64_t)pr[i].match_off <= h.total_matches &&
                 (uint64_t)pr[i].m <= h.total_matches - (uint64_t)pr[i].match_off;
    }
    if (!ok) {
        std::free(block);
        return GMS_ERR_IO;
    }
    d->n_frames = (int32_t)h.n_frames;
    d->desc_kind = (int32_t)h.desc_kind - 1;
    d->n_pairs = (int64_t)h.n_pairs;
    d->total_matches = (int64_t)h.total_matches;
    d->wh = (int32_t*)a_wh;
    d->frame_off = (int64_t*)a_off;
    d->keypoints = (gms_keypoint*)a_kp;
    d->descriptors = row ? a_desc : nullptr;
    d->pairs = (gms_pair*)a_pairs;
    d->matches = (gms_dmatch*)a_m;
    d->owner = block;
    return GMS_OK;
}

void gms_dataset_free(gms_dataset* d)
{
    if (d && d->owner) std::free(d->owner);
    if (d) std::memset(d, 0, sizeof *d);
}

}  // extern "C"
