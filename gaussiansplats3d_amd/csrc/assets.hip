// assets.hip — native asset readers (host code only): INRIA-v1 .ply and .ksplat -> the arrays the render / sort seams
// consume.  Restates, never copies:
//   PLY header        /root/reference/src/loaders/ply/PlyParserUtils.js:31-165 (decodeSectionHeader, SH field mapping),
//                     INRIAV1PlyParser.js:20-47 (fields read)
//   PLY row -> splat  INRIAV1PlyParser.js:114-209 (exp scale, sigmoid opacity, floor/clamp colours, quaternion normalise)
//                     + SplatBuffer.writeSplatDataToSectionBuffer :1056-1113 (level-0 row: second normalise, fp32 stores)
//   .ksplat layout    SplatBuffer.js:108-163 (rows per compression level), :819-848 (header), :877-941 (section headers),
//                     :199-219 (bucket of a splat), :221-246 (centre decode)
//   arrays            SplatBuffer.fillSplatCenterArray / fillSplatColorArray :551-575 / fillSplatCovarianceArray :517-549 +
//                     computeCovariance :440-486 (three.js Matrix3/4 arithmetic in double) / fillSphericalHarmonicsArray
//                     :577-734 WITHOUT a scene transform (identity scenes / dynamicMode), SH target level =
//                     max(1, buffer level) (SplatMesh.js:1064-1066)
// A PLY is first laid out as the level-0 section the reference would build from it (file order, i.e. the reference's
// `optimizeSplatData: false`), so every fill routine reads one format.
#include <algorithm>
#include <math.h>
#include <string>

#include "gs_internal.hpp"

namespace {

// THREE.DataUtils.toHalfFloat (three r160): clamp to +-65504, then the base/shift tables: the mantissa is TRUNCATED
uint16_t to_half_three(double value) {
    float v = (float)value;
    if (v > 65504.0f) v = 65504.0f;
    if (v < -65504.0f) v = -65504.0f;
    uint32_t f;
    memcpy(&f, &v, 4);
    const uint32_t sign = (f >> 16) & 0x8000u;
    const int e = (int)((f >> 23) & 0xFFu) - 127;
    const uint32_t m = f & 0x007FFFFFu;
    uint32_t out;
    if (e < -24) out = 0;
    else if (e < -14) out = (0x0400u >> (-e - 14)) + (m >> (-e - 1));
    else if (e <= 15) out = ((uint32_t)(e + 15) << 10) + (m >> 13);
    else if (e < 128) out = 0x7C00u;
    else out = 0x7C00u + (m >> 13);
    return (uint16_t)(out | sign);
}

double from_half(uint16_t h) {                         // exact
    const uint32_t sign = h & 0x8000u, e = (h >> 10) & 31u, m = h & 1023u;
    double v;
    if (e == 0) v = ldexp((double)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexp((double)(m | 1024u), (int)e - 25);
    return sign ? -v : v;
}

// Util.js clamp = Math.max(Math.min(v, hi), lo): JS min / max PROPAGATE NaN (C's fmin / fmax drop it)
double clampd(double v, double lo, double hi) {
    if (v != v) return v;
    return v > hi ? hi : (v < lo ? lo : v);
}

uint8_t to_uint8_range(double v, double lo, double hi) {                            // SplatBuffer.js:22-26
    v = clampd(v, lo, hi);
    const double r = clampd(floor((v - lo) / (hi - lo) * 255.0), 0.0, 255.0);
    return r == r ? (uint8_t)r : (uint8_t)0;                                        // a NaN stored into a Uint8Array is 0
}

uint8_t clamped_u8(double v) {                         // Uint8ClampedArray store: NaN -> 0, round half to even
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)nearbyint(v);
}

struct Section {
    uint32_t splat_count, max_splat_count, bucket_size, bucket_count, full_buckets, partial_buckets, sh_degree;
    uint32_t bytes_per_splat, scale_range;
    double half_block, scale_factor;
    size_t base, buckets_base, data_base;
    uint32_t count_offset;
    uint32_t bucket_storage;
    std::vector<uint32_t> partial_end;     // cumulative end (in section-local splats) of every partial bucket
};

constexpr size_t KS_HEADER = 4096, KS_SECTION_HEADER = 1024;
const uint32_t CENTER_BYTES[3] = {12, 6, 6}, SCALE_BYTES[3] = {12, 6, 6}, ROT_BYTES[3] = {16, 8, 8}, SH_BYTES_PER[3] = {4, 2, 1};
uint32_t sh_components(uint32_t degree) { return degree == 0 ? 0u : (degree == 1 ? 9u : 24u); }

}  // namespace

struct gs_asset {
    std::vector<uint8_t> buf;              // a .ksplat image (for a PLY: the level-0 section built from it)
    uint32_t level = 0, splat_count = 0, sh_degree = 0;
    float scene_center[3] = {0, 0, 0};
    double sh_min = -1.5, sh_max = 1.5;
    std::vector<Section> sections;
    std::vector<uint32_t> section_of;      // per splat

    template <class T>
    T rd(size_t off) const {
        T v;
        memcpy(&v, buf.data() + off, sizeof(T));
        return v;
    }
};

namespace {

int parse_ksplat(gs_asset* a) {
    const size_t n = a->buf.size();
    GS_REQUIRE(n >= KS_HEADER, ".ksplat: shorter than its 4096-byte header");
    const uint32_t max_sections = a->rd<uint32_t>(4), max_splats = a->rd<uint32_t>(12);
    a->level = a->rd<uint16_t>(20);
    GS_REQUIRE(a->level <= 2, ".ksplat: unknown compression level");
    for (int k = 0; k < 3; k++) a->scene_center[k] = a->rd<float>(24 + 4 * k);
    const float mn = a->rd<float>(36), mx = a->rd<float>(40);
    a->sh_min = mn != 0.0f ? (double)mn : -1.5;            // `|| -DefaultSphericalHarmonics8BitCompressionHalfRange`
    a->sh_max = mx != 0.0f ? (double)mx : 1.5;
    GS_REQUIRE(KS_HEADER + (size_t)max_sections * KS_SECTION_HEADER <= n, ".ksplat: section headers exceed the file");
    size_t base = KS_HEADER + (size_t)max_sections * KS_SECTION_HEADER;
    uint32_t count_offset = 0;
    uint32_t min_degree = 0;
    for (uint32_t s = 0; s < max_sections; s++) {
        const size_t h = KS_HEADER + (size_t)s * KS_SECTION_HEADER;
        Section sec = {};
        sec.max_splat_count = a->rd<uint32_t>(h + 4);
        sec.splat_count = sec.max_splat_count;             // secLoadedCountsToMax
        sec.bucket_size = a->rd<uint32_t>(h + 8);
        sec.bucket_count = a->rd<uint32_t>(h + 12);
        const float block = a->rd<float>(h + 16);
        sec.half_block = (double)block / 2.0;
        const uint32_t bucket_storage = a->rd<uint16_t>(h + 20);
        sec.bucket_storage = bucket_storage;
        const uint32_t range = a->rd<uint32_t>(h + 24);
        sec.scale_range = range ? range : (a->level == 0 ? 1u : 32767u);
        sec.scale_factor = sec.half_block / (double)sec.scale_range;
        sec.full_buckets = a->rd<uint32_t>(h + 32);
        sec.partial_buckets = a->rd<uint32_t>(h + 36);
        sec.sh_degree = a->rd<uint16_t>(h + 40);
        GS_REQUIRE(sec.sh_degree <= 2, ".ksplat: spherical harmonics degree > 2");
        sec.bytes_per_splat = CENTER_BYTES[a->level] + SCALE_BYTES[a->level] + ROT_BYTES[a->level] + 4u +
                              SH_BYTES_PER[a->level] * sh_components(sec.sh_degree);
        const size_t meta = (size_t)sec.partial_buckets * 4, buckets = (size_t)bucket_storage * sec.bucket_count + meta;
        sec.base = base;
        sec.buckets_base = base + meta;
        sec.data_base = base + buckets;
        sec.count_offset = count_offset;
        const size_t end = sec.data_base + (size_t)sec.bytes_per_splat * sec.max_splat_count;
        GS_REQUIRE(end <= n, ".ksplat: section data exceeds the file");
        if (a->level > 0 && sec.max_splat_count > 0) {
            // Bucket tables are only read for compressed centres (SplatBuffer.js:199-246).  The reference is memory-safe
            // JavaScript; here every index into the tables is proven in range before gs_asset_fill reads through them.
            GS_REQUIRE(bucket_storage >= 12, ".ksplat: bucket storage below the 12 bytes of a bucket centre");
            GS_REQUIRE(sec.bucket_size > 0, ".ksplat: bucket size 0 in a compressed section");
            GS_REQUIRE((uint64_t)sec.full_buckets + sec.partial_buckets <= sec.bucket_count,
                       ".ksplat: more full + partial buckets than the section stores");
            uint64_t covered = (uint64_t)sec.full_buckets * sec.bucket_size;
            GS_REQUIRE(covered <= 0xFFFFFFFFull, ".ksplat: full buckets x bucket size overflows 32 bits");   // bucket_index's span
            sec.partial_end.reserve(sec.partial_buckets);
            for (uint32_t p = 0; p < sec.partial_buckets; p++) {
                covered += a->rd<uint32_t>(sec.base + 4 * (size_t)p);
                GS_REQUIRE(covered <= 0xFFFFFFFFull, ".ksplat: partial bucket lengths overflow");
                sec.partial_end.push_back((uint32_t)covered);
            }
            GS_REQUIRE(covered >= sec.max_splat_count, ".ksplat: the buckets do not cover every splat of the section");
        }
        base = end;
        count_offset += sec.max_splat_count;
        if (s == 0 || sec.sh_degree < min_degree) min_degree = sec.sh_degree;   // getMinSphericalHarmonicsDegree
        a->sections.push_back(sec);
    }
    GS_REQUIRE(count_offset == max_splats || max_sections == 0 || count_offset >= max_splats, ".ksplat: splat counts disagree");
    a->splat_count = count_offset < max_splats ? count_offset : max_splats;
    a->sh_degree = min_degree;
    a->section_of.resize(count_offset);
    for (uint32_t s = 0; s < a->sections.size(); s++)
        for (uint32_t j = 0; j < a->sections[s].max_splat_count; j++) a->section_of[a->sections[s].count_offset + j] = s;
    return GS_OK;
}

// ---- PLY ------------------------------------------------------------------------------------------
enum FieldType { T_DOUBLE, T_INT, T_UINT, T_FLOAT, T_SHORT, T_USHORT, T_UCHAR, T_UNKNOWN };
int field_size(FieldType t) {
    switch (t) {
        case T_DOUBLE: return 8;
        case T_INT: case T_UINT: case T_FLOAT: return 4;
        case T_SHORT: case T_USHORT: return 2;
        case T_UCHAR: return 1;
        default: return -1;
    }
}
FieldType field_type(const std::string& s) {
    if (s == "double") return T_DOUBLE;
    if (s == "int") return T_INT;
    if (s == "uint") return T_UINT;
    if (s == "float") return T_FLOAT;
    if (s == "short") return T_SHORT;
    if (s == "ushort") return T_USHORT;
    if (s == "uchar") return T_UCHAR;
    return T_UNKNOWN;
}

struct PlyField {
    bool present = false;
    FieldType type = T_UNKNOWN;
    size_t offset = 0;
};

// PlyParserUtils.readVertex for one field: false = `undefined`
bool read_field(const uint8_t* row, const PlyField& f, double* out) {
    if (!f.present) return false;
    const uint8_t* p = row + f.offset;
    switch (f.type) {
        case T_FLOAT: { float v; memcpy(&v, p, 4); *out = v; return true; }
        case T_SHORT: { int16_t v; memcpy(&v, p, 2); *out = v; return true; }
        case T_USHORT: { uint16_t v; memcpy(&v, p, 2); *out = v; return true; }
        case T_INT: { int32_t v; memcpy(&v, p, 4); *out = v; return true; }
        case T_UINT: { uint32_t v; memcpy(&v, p, 4); *out = v; return true; }
        case T_UCHAR: *out = (double)*p / 255.0; return true;                          // normalize = true
        default: return false;                                                         // doubles are never read (:281-301)
    }
}

std::string trim(const std::string& s) {
    size_t b = 0, e = s.size();
    while (b < e && isspace((unsigned char)s[b])) b++;
    while (e > b && isspace((unsigned char)s[e - 1])) e--;
    return s.substr(b, e - b);
}

int parse_ply(gs_asset* a, const uint8_t* data, size_t bytes, uint32_t want_degree) {
    const std::string token = "end_header";
    const std::string head((const char*)data, bytes < (1u << 20) ? bytes : (1u << 20));
    const size_t tok = head.find(token);
    GS_REQUIRE(tok != std::string::npos, "PLY: end_header not found");
    const size_t header_bytes = tok + token.size() + 1;                                // INRIAV1PlyParser.js:53
    // decodeSectionHeader: first `element` section only
    std::vector<std::pair<std::string, FieldType>> fields;
    uint32_t vertex_count = 0;
    bool in_section = false;
    size_t pos = 0;
    while (pos < tok + token.size()) {
        size_t nl = head.find('\n', pos);
        if (nl == std::string::npos) nl = head.size();
        const std::string line = trim(head.substr(pos, nl - pos));
        pos = nl + 1;
        if (line.rfind("element", 0) == 0) {
            if (in_section) break;
            in_section = true;
            size_t p1 = line.find_first_not_of(' ', 7);                                // components split on ' '
            size_t p2 = line.find(' ', p1);
            size_t p3 = p2 == std::string::npos ? p2 : line.find_first_not_of(' ', p2);
            if (p3 != std::string::npos) vertex_count = (uint32_t)strtoul(line.c_str() + p3, nullptr, 10);
        } else if (line.rfind("property", 0) == 0) {
            char w0[64], w1[64], w2[128];
            if (sscanf(line.c_str(), "%63[A-Za-z0-9_] %63[A-Za-z0-9_] %127[A-Za-z0-9_]", w0, w1, w2) == 3)     // /(\\w+)\\s+(\\w+)\\s+(\\w+)/
                fields.push_back({w2, field_type(w1)});
        }
        if (line == token) break;
    }
    size_t bytes_per_vertex = 0;
    uint32_t f_rest = 0;
    for (auto& f : fields) {
        GS_REQUIRE(field_size(f.second) > 0, "PLY: property type the reference does not size (bytesPerVertex would be NaN)");
        if (f.first.rfind("f_rest", 0) == 0) f_rest++;
    }
    // INRIAV1PlyParser.decodeHeaderLines: how many f_rest_* fields enter the name->id map
    const uint32_t sh_to_read = f_rest >= 45 ? 45u : (f_rest >= 24 ? 24u : (f_rest >= 9 ? 9u : 0u));
    auto mapped = [&](const std::string& name) {
        static const char* base[] = {"scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3", "x", "y", "z", "f_dc_0",
                                     "f_dc_1", "f_dc_2", "opacity", "red", "green", "blue", "f_rest_0"};
        for (const char* b : base)
            if (name == b) return true;
        if (name.rfind("f_rest_", 0) == 0) {
            const unsigned long k = strtoul(name.c_str() + 7, nullptr, 10);
            return k >= 1 && k + 1 <= sh_to_read && name == "f_rest_" + std::to_string(k);
        }
        return false;
    };
    auto find = [&](const std::string& name) {
        PlyField r;
        size_t off = 0;
        for (auto& f : fields) {
            if (f.first == name && mapped(name)) {                                     // later duplicates overwrite: keep the last
                r.present = true;
                r.type = f.second;
                r.offset = off;
            }
            off += (size_t)field_size(f.second);
        }
        return r;
    };
    for (auto& f : fields) bytes_per_vertex += (size_t)field_size(f.second);
    const double cpc_d = (double)f_rest / 3.0;                                         // coefficientsPerChannel (may be fractional)
    uint32_t degree = 0;
    if (cpc_d >= 3) degree = 1;
    if (cpc_d >= 8) degree = 2;
    const uint32_t out_degree = want_degree < degree ? want_degree : degree;
    GS_REQUIRE(header_bytes + bytes_per_vertex * (size_t)vertex_count <= bytes, "PLY: vertex data exceeds the file");
    PlyField F_scale[3], F_rot[4], F_pos[3], F_dc[3], F_op, F_rgb[3], F_rest0, F_d1[9], F_d2[15];
    for (int k = 0; k < 3; k++) F_scale[k] = find("scale_" + std::to_string(k));
    for (int k = 0; k < 4; k++) F_rot[k] = find("rot_" + std::to_string(k));
    F_pos[0] = find("x"); F_pos[1] = find("y"); F_pos[2] = find("z");
    for (int k = 0; k < 3; k++) F_dc[k] = find("f_dc_" + std::to_string(k));
    F_op = find("opacity");
    F_rgb[0] = find("red"); F_rgb[1] = find("green"); F_rgb[2] = find("blue");
    F_rest0 = find("f_rest_0");
    // decodeSphericalHarmonicsFromSectionHeader: 'f_rest_' + (i + cpc*rgb [+ 3]); JS number -> string
    auto rest_name = [&](double k) {
        char t[64];
        if (k == floor(k)) snprintf(t, sizeof(t), "f_rest_%.0f", k);
        else snprintf(t, sizeof(t), "f_rest_%.17g", k);
        return std::string(t);
    };
    for (int rgb = 0; rgb < 3; rgb++) {
        if (degree >= 1) for (int i = 0; i < 3; i++) F_d1[3 * rgb + i] = find(rest_name(i + cpc_d * rgb));
        if (degree >= 2) for (int i = 0; i < 5; i++) F_d2[5 * rgb + i] = find(rest_name(i + cpc_d * rgb + 3));
    }

    // level-0 .ksplat image with one section (SplatBuffer.preallocateUncompressed :1401-1433)
    const uint32_t ncomp = sh_components(out_degree), bps = 44u + 4u * ncomp;
    a->buf.assign(KS_HEADER + KS_SECTION_HEADER + (size_t)bps * vertex_count, 0);
    uint8_t* B = a->buf.data();
    auto W32 = [&](size_t off, uint32_t v) { memcpy(B + off, &v, 4); };
    auto W16 = [&](size_t off, uint16_t v) { memcpy(B + off, &v, 2); };
    auto WF = [&](size_t off, float v) { memcpy(B + off, &v, 4); };
    B[0] = 0; B[1] = 1;
    W32(4, 1); W32(8, 1); W32(12, vertex_count); W32(16, vertex_count); W16(20, 0);
    WF(36, -1.5f); WF(40, 1.5f);
    W32(KS_HEADER + 0, vertex_count); W32(KS_HEADER + 4, vertex_count); W16(KS_HEADER + 40, (uint16_t)out_degree);
    const uint8_t* rows = data + header_bytes;
    for (uint32_t i = 0; i < vertex_count; i++) {
        const uint8_t* row = rows + (size_t)i * bytes_per_vertex;
        const size_t o = KS_HEADER + KS_SECTION_HEADER + (size_t)i * bps;
        double v, s3[3], r4[4] = {NAN, NAN, NAN, NAN}, c3[3] = {NAN, NAN, NAN}, col[3], op = 0.0;   // createSplat() starts every field at 0
        // INRIAV1PlyParser.js:148-156
        if (read_field(row, F_scale[0], &v)) {
            for (int k = 0; k < 3; k++) { s3[k] = NAN; if (read_field(row, F_scale[k], &v)) s3[k] = exp(v); }
        } else {
            s3[0] = s3[1] = s3[2] = 0.01;
        }
        // :158-172
        if (read_field(row, F_dc[0], &v)) {
            const double SH_C0 = 0.28209479177387814;
            for (int k = 0; k < 3; k++) { col[k] = NAN; if (read_field(row, F_dc[k], &v)) col[k] = (0.5 + SH_C0 * v) * 255; }
        } else if (read_field(row, F_rgb[0], &v)) {
            for (int k = 0; k < 3; k++) { col[k] = NAN; if (read_field(row, F_rgb[k], &v)) col[k] = v * 255; }
        } else {
            col[0] = col[1] = col[2] = 0;
        }
        if (read_field(row, F_op, &v)) op = (1 / (1 + exp(-v))) * 255;                 // :174-176
        for (int k = 0; k < 3; k++) col[k] = clampd(floor(col[k]), 0, 255);            // :178-181
        op = clampd(floor(op), 0, 255);
        // :196-202 Quaternion.set(rot_0..3).normalize(), then the second normalize of writeSplatDataToSectionBuffer :1084-1086
        for (int k = 0; k < 4; k++) if (read_field(row, F_rot[k], &v)) r4[k] = v;
        for (int pass = 0; pass < 2; pass++) {
            double l = sqrt(r4[0] * r4[0] + r4[1] * r4[1] + r4[2] * r4[2] + r4[3] * r4[3]);   // x*x + y*y + z*z + w*w
            if (l == 0) { r4[0] = r4[1] = r4[2] = 0; r4[3] = 1; }
            else { l = 1 / l; for (int k = 0; k < 4; k++) r4[k] = r4[k] * l; }
        }
        for (int k = 0; k < 3; k++) if (read_field(row, F_pos[k], &v)) c3[k] = v;
        for (int k = 0; k < 3; k++) WF(o + 4 * k, (float)c3[k]);
        for (int k = 0; k < 3; k++) WF(o + 12 + 4 * k, (float)(s3[k] == s3[k] ? s3[k] : 0.0));   // `|| 0`
        for (int k = 0; k < 4; k++) WF(o + 24 + 4 * k, (float)r4[k]);
        B[o + 40] = clamped_u8(col[0]); B[o + 41] = clamped_u8(col[1]); B[o + 42] = clamped_u8(col[2]);
        B[o + 43] = clamped_u8(op);
        if (out_degree >= 1) {                                                         // :183-194
            const bool have = read_field(row, F_rest0, &v);
            for (int s = 0; s < 9; s++) {
                double c = 0;
                if (have && read_field(row, F_d1[s], &v)) c = v;
                WF(o + 44 + 4 * s, (float)c);
            }
            if (out_degree >= 2)
                for (int s = 0; s < 15; s++) {
                    double c = 0;
                    if (have && read_field(row, F_d2[s], &v)) c = v;
                    WF(o + 44 + 36 + 4 * s, (float)c);
                }
        }
    }
    return parse_ksplat(a);
}

// SplatBuffer.js:199-219: full buckets first, then the partial ones by their stored lengths.  parse_ksplat proved that the
// tables cover every splat, so the result is always < bucket_count.
uint32_t bucket_index(const gs_asset*, const Section& sec, uint32_t local) {
    const uint32_t full_span = sec.full_buckets * sec.bucket_size;
    if (local < full_span) return local / sec.bucket_size;
    const auto it = std::upper_bound(sec.partial_end.begin(), sec.partial_end.end(), local);
    const uint32_t b = sec.full_buckets + (uint32_t)(it - sec.partial_end.begin());
    return b < sec.bucket_count ? b : sec.bucket_count - 1u;       // unreachable after parse_ksplat's checks; never past the table
}

double comp(const gs_asset* a, size_t row, uint32_t index, bool sh) {                   // dataViewFloatForCompressionLevel + toUncompressedFloat
    if (a->level == 0) return a->rd<float>(row + 4 * (size_t)index);
    if (a->level == 1 || !sh) return from_half(a->rd<uint16_t>(row + 2 * (size_t)index));
    return (double)a->rd<uint8_t>(row + index) / 255 * (a->sh_max - a->sh_min) + a->sh_min;
}

}  // namespace

extern "C" {

int gs_asset_open(const void* data, uint64_t bytes, uint32_t format, uint32_t max_sh_degree, gs_asset** out) {
    GS_REQUIRE(data && out, "data / out == NULL");
    *out = nullptr;
    gs_asset* a = new (std::nothrow) gs_asset();
    if (!a) return GS_ERR_NOMEM;
    int st;
    try {
        if (format == GS_ASSET_PLY) {
            st = parse_ply(a, (const uint8_t*)data, (size_t)bytes, max_sh_degree);
        } else if (format == GS_ASSET_KSPLAT) {
            a->buf.assign((const uint8_t*)data, (const uint8_t*)data + bytes);
            st = parse_ksplat(a);
            if (st == GS_OK && a->sh_degree > max_sh_degree) a->sh_degree = max_sh_degree;
        } else {
            gs_set_error("invalid argument: unknown asset format");
            st = GS_ERR_INVALID;
        }
    } catch (const std::bad_alloc&) {
        gs_set_error("out of host memory while reading the asset");
        st = GS_ERR_NOMEM;
    }
    if (st != GS_OK) {
        delete a;
        return st;
    }
    *out = a;
    return GS_OK;
}

void gs_asset_close(gs_asset* a) { delete a; }

int gs_asset_get_info(gs_asset* a, gs_asset_info* info) {
    GS_REQUIRE(a && info, "asset / info == NULL");
    info->splat_count = a->splat_count;
    info->sh_degree = a->sh_degree;
    info->compression_level = a->level;
    info->sh_level = a->level < 1 ? 1u : a->level;          // getTargetSphericalHarmonicsCompressionLevel
    for (int k = 0; k < 3; k++) info->scene_center[k] = a->scene_center[k];
    info->sh_min = (float)a->sh_min;
    info->sh_max = (float)a->sh_max;
    return GS_OK;
}

int gs_asset_fill(gs_asset* a, uint32_t min_alpha, float* centers, float* cov_f32, uint16_t* cov_f16, uint8_t* rgba,
                  uint16_t* sh_f16, uint8_t* sh_u8, float* scales, float* rotations) {
    GS_REQUIRE(a != nullptr, "asset == NULL");
    GS_REQUIRE(!(sh_f16 && sh_u8), "pass sh_f16 (compression level <= 1) or sh_u8 (level 2), not both");
    GS_REQUIRE(!sh_u8 || a->level == 2, "sh_u8 output needs a compression level 2 file (SplatMesh.js:1064-1066)");
    GS_REQUIRE(!sh_f16 || a->level <= 1, "a level 2 file keeps its SH as uint8: ask for sh_u8");
    const uint32_t ncomp = sh_components(a->sh_degree);
    for (uint32_t i = 0; i < a->splat_count; i++) {
        const Section& sec = a->sections[a->section_of[i]];
        const uint32_t local = i - sec.count_offset;
        const size_t row = sec.data_base + (size_t)sec.bytes_per_splat * local;
        if (centers) {                                                                 // getSplatCenter :221-246
            if (a->level == 0) {
                for (int k = 0; k < 3; k++) centers[3 * (size_t)i + k] = a->rd<float>(row + 4 * k);
            } else {
                const uint32_t b = bucket_index(a, sec, local);
                for (int k = 0; k < 3; k++) {
                    const double x = a->rd<uint16_t>(row + 2 * k);
                    const double bc = a->rd<float>(sec.buckets_base + (size_t)sec.bucket_storage * b + 4 * k);
                    centers[3 * (size_t)i + k] = (float)((x - (double)sec.scale_range) * sec.scale_factor + bc);
                }
            }
        }
        const size_t srow = row + CENTER_BYTES[a->level];
        if (cov_f32 || cov_f16 || scales || rotations) {
            const double sx = comp(a, srow, 0, false), sy = comp(a, srow, 1, false), sz = comp(a, srow, 2, false);
            // rotation.set(x = f4, y = f5, z = f6, w = f3): NOT normalised (:539-542)
            const double w = comp(a, srow, 3, false), x = comp(a, srow, 4, false), y = comp(a, srow, 5, false), z = comp(a, srow, 6, false);
            if (scales) { scales[3 * (size_t)i] = (float)sx; scales[3 * (size_t)i + 1] = (float)sy; scales[3 * (size_t)i + 2] = (float)sz; }
            if (rotations) {
                // fillSplatScaleRotationArray (SplatBuffer.js:407-424): Quaternion.normalize, then ensurePositiveW
                double q[4] = {x, y, z, w};
                double l = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
                if (l == 0) { q[0] = q[1] = q[2] = 0; q[3] = 1; }
                else { l = 1 / l; for (int k = 0; k < 4; k++) q[k] = q[k] * l; }
                const double flip = q[3] < 0 ? -1 : 1;
                for (int k = 0; k < 4; k++) rotations[4 * (size_t)i + k] = (float)(q[k] * flip);
            }
            if (cov_f32 || cov_f16) {
                // Matrix4.makeRotationFromQuaternion = compose(zero, q, one) (three r160)
                const double x2 = x + x, y2 = y + y, z2 = z + z;
                const double xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2;
                const double wx = w * x2, wy = w * y2, wz = w * z2;
                const double R[3][3] = {{(1 - (yy + zz)) * 1, (xy - wz) * 1, (xz + wy) * 1},
                                        {(xy + wz) * 1, (1 - (xx + zz)) * 1, (yz - wx) * 1},
                                        {(xz - wy) * 1, (yz + wx) * 1, (1 - (xx + yy)) * 1}};
                // covarianceMatrix = R * S (Matrix3.multiplyMatrices: a_i1*b_1j + a_i2*b_2j + a_i3*b_3j)
                const double S[3][3] = {{sx, 0, 0}, {0, sy, 0}, {0, 0, sz}};
                double M[3][3], Cm[3][3];
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++) M[r][c] = R[r][0] * S[0][c] + R[r][1] * S[1][c] + R[r][2] * S[2][c];
                // transformedCovariance = M * M^T
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++) Cm[r][c] = M[r][0] * M[c][0] + M[r][1] * M[c][1] + M[r][2] * M[c][2];
                const double e[6] = {Cm[0][0], Cm[0][1], Cm[0][2], Cm[1][1], Cm[1][2], Cm[2][2]};   // elements 0,3,6,4,7,8
                for (int k = 0; k < 6; k++) {
                    if (cov_f32) cov_f32[6 * (size_t)i + k] = (float)e[k];
                    if (cov_f16) cov_f16[6 * (size_t)i + k] = to_half_three(e[k]);
                }
            }
        }
        if (rgba) {                                                                    // fillSplatColorArray :551-575
            const size_t crow = srow + SCALE_BYTES[a->level] + ROT_BYTES[a->level];
            for (int k = 0; k < 3; k++) rgba[4 * (size_t)i + k] = a->buf[crow + k];
            const uint8_t alpha = a->buf[crow + 3];
            rgba[4 * (size_t)i + 3] = alpha >= min_alpha ? alpha : 0;
        }
        if ((sh_f16 || sh_u8) && ncomp) {                                              // fillSphericalHarmonicsArray, no transform
            const size_t hrow = srow + SCALE_BYTES[a->level] + ROT_BYTES[a->level] + 4u;
            auto emit = [&](uint32_t dst, uint32_t src) {
                if (sh_u8) { sh_u8[(size_t)ncomp * i + dst] = a->rd<uint8_t>(hrow + src); return; }
                sh_f16[(size_t)ncomp * i + dst] = a->level == 0 ? to_half_three(a->rd<float>(hrow + 4 * (size_t)src))
                                                                : a->rd<uint16_t>(hrow + 2 * (size_t)src);
            };
            for (uint32_t c = 0; c < 3; c++)                                           // set3FromArray(stride 3, base c)
                for (uint32_t ch = 0; ch < 3; ch++) emit(3 * c + ch, c + 3 * ch);
            if (a->sh_degree >= 2)
                for (uint32_t c = 0; c < 5; c++)                                       // set3FromArray(stride 5, base 9 + c)
                    for (uint32_t ch = 0; ch < 3; ch++) emit(9 + 3 * c + ch, 9 + c + 5 * ch);
        }
    }
    return GS_OK;
}

}  // extern "C"
