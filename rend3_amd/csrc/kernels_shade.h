// kernels_shade.h -- deferred PBR resolve of the visibility buffer (K6 shading: opaque.wgsl VS + FS, brdf.wgsl, pcf.wgsl),
// the transparent pass's ordered blend (row N3) and the tonemap blit (K7).  Built as its own translation unit
// (shade.hip); r3n.hip sees only the argument structures and the launchers declared at the end of this file
// (R3N_SHADE_DECL_ONLY).
//
// Reference behaviour restated (file:line):
//   opaque.wgsl:91-135 (VS), :203-551 (FS), math/brdf.wgsl, shadow/pcf.wgsl
//   blit.wgsl:22-31, tonemapping.rs:44 (Rgba8UnormSrgb target => exact sRGB OETF)
#pragma once
#include <type_traits>

#include "device_math.h"
#include "texture.h"

// The resolve's constants -- fixed by measurement, NOT build options (an experiment is a patch: profiles/patches/).
#define R3N_TEX_OCC 5  // min waves per SIMD asked of the textured record-based resolve (launch bound)
#define R3N_MS_OCC 4   // the same for the multisampled record-based resolve and its edge pass.  5 is 9 % faster (shade 857 vs 935 us on
                       // the bench scene at four samples) but costs the textured instantiations 480 spilled vector registers, 80 spilled
                       // scalar ones and 1.1 KB of scratch per lane (4: 100 B) -- and round 6 met a build of exactly these kernels that
                       // never returned at 5 and ran at 4 with the same source (profiles/r06_native_hang.md): spill code at the
                       // register limit is not worth 9 % of a non-headline mode
// Measured and NOT kept (code in the history, commit 79bedf3; numbers in profiles/r0N_summary.md): resolve tiles in contiguous
// per-XCD bands (shade 595 -> 593 us, frame 1.184 -> 1.193 ms); the directional lights in groups of two / four whose shadow
// texels are in flight together (461.0 vs 460.6 us; four: 624 us at three waves per SIMD); round 5: the lookups of every light
// issued in front of the texture chain (460.8 -> 474.0 / 493.6 us, profiles/patches/r05_resolve_early_shadow_lookups.patch);
// the parts of the fragment stage compiled out one by one to cost them (profiles/r04_summary.md section 6b).

// ------------------------------------------------------------------------------------------------ K6 resolve
struct TriRecord;
struct ViewLights;
struct ShadeArgs {
    const unsigned long long *vis;
    uint32_t width, height, row_begin, row_end;
    const r3n_frame_uniforms496 *fu;
    const r3n_camera_header240 *hdr;
    const r3n_object128 *objects;
    const uint32_t *mesh;
    const r3n_baked128 *baked;
    const r3n_material208 *materials;
    uint32_t n_materials;
    const uint32_t *tri_base;
    const uint32_t *slot_table;  // slot_table[b] = object owning canonical slot b << R3N_SLOT_TABLE_SHIFT
    uint32_t slot_table_size;
    const uint8_t *dir_buf;    // count @0, records @16
    const uint8_t *point_buf;  // count @0, records @16
    const float *atlas;
    uint32_t atlas_w, atlas_h;
    float clear[4];
    ushort4 *hdr_out;          // Rgba16Float
    uchar4 *ldr_out;           // Rgba8UnormSrgb: the tonemap blit fused into the resolve (one HDR round trip less)
    const unsigned char *srgb_lut;
    bool out_bgr;              // Bgra8* output target
    TextureArgs tex;
    ushort4 *samples_out;      // S == 4 and a transparent pass follows: the per-sample colours (else null)
    TriRecord *tri_rec;        // S == 1: per-triangle vertex-stage records by canonical slot (else null)
    unsigned char *seen;       // ... and which slots own a pixel this frame
    uint32_t total_tris;
    // S == 4, split resolve: pixels whose samples belong to more than one triangle hand their extra triangles to a
    // second, dense pass (R3N_EDGEQ sub-lists of pixel << 3 | leader sample << 1 | last-entry-of-the-pixel)
    uint32_t *edge_list, *edge_count;
    uint32_t edge_capacity;    // entries per sub-list
    // single-sample record-based resolve, specialised per material class (see R3N_FEAT_* below)
    const ViewLights *view_lights;  // the frame's light lists in view space (k_stage_view_lights), or null: every workgroup stages its own
    const uint32_t *material_feat;  // per material: feature bits, computed by the host when the record is written
    uint32_t variants;              // bit v set: variant v of the resolve runs this frame (more than one bit: tiles are classified)
    uint32_t resolve_lds;           // dynamic LDS bytes asked by the single-sample resolve launches (unused by the kernel: a cap on its resident workgroups, r3n.hip Tune)
};
#define R3N_EDGEQ 32u

// ---- material classes of the resolve -------------------------------------------------------------------------------------
// opaque.wgsl:203-424 branches on fifteen flag bits and ten texture slots at run time (material.wgsl:1-19).  A material's FEATURES
// are the branches it can take; a resolve variant instantiated for a feature set K contains only the code of K's features and
// shades exactly like the general kernel every material whose features are a subset of K (the tests of the remaining features
// stay run-time tests; the removed ones would have been false).  The variants form a chain PLAIN < ALBEDO < PBR3 < ALL, so a
// 16x16 tile is shaded by the variant of the OR of its pixels' features, and a world whose materials all map to one variant
// needs no classification at all.
#define R3N_FEAT_TEX_ALBEDO   0x001u  // slot 0 bound
#define R3N_FEAT_TEX_NORMAL   0x002u  // slot 1 bound
#define R3N_FEAT_TEX_AOMR     0x004u  // slot 2 bound
#define R3N_FEAT_TEX_AOMR_X   0x008u  // slot 3 or 9 bound (the split layouts' other maps)
#define R3N_FEAT_TEX_MISC     0x010u  // slot 4, 5, 6 or 7 bound (reflectance, clear coat, clear-coat roughness, emissive)
#define R3N_FEAT_VCOLOR       0x020u  // ALBEDO_BLEND: the vertex colour multiplies the albedo
#define R3N_FEAT_UNLIT        0x040u
#define R3N_FEAT_NEAREST      0x080u  // nearest sampler
#define R3N_FEAT_AOMR_SPLIT   0x100u  // slot 2 / 3 / 9 bound and the layout is not AOMR_COMBINED
#define R3N_FEAT_CLEARCOAT    0x200u  // clear_coat factor != 0
#define R3N_FEAT_NORMAL_FLAGS 0x400u  // BICOMPONENT / SWIZZLED / YDOWN normal map
#define R3N_FEAT_ALBEDO_OFF   0x800u  // ALBEDO_ACTIVE clear
#define R3N_FEAT_TEX_GENERAL  0x1000u // binds a texture outside the sampler's short path (set by the host's census, r3n.hip:
                                      // extent not a power of two, float pool texels, pool beyond 2^30 texels, id out of range)
#define R3N_FEAT_ALL          0x1FFFu
#define R3N_BATCH_OCC 4  // waves per SIMD asked of the PBR class's kernel: its batched sampler holds 24 texels in flight (128 VGPRs)
#define R3N_CLS_PLAIN  0u
#define R3N_CLS_ALBEDO (R3N_FEAT_TEX_ALBEDO)
#define R3N_CLS_PBR3   (R3N_FEAT_TEX_ALBEDO | R3N_FEAT_TEX_NORMAL | R3N_FEAT_TEX_AOMR)
#define R3N_CLS_ALL    R3N_FEAT_ALL
#define R3N_VARIANTS 4u  // 0 PLAIN, 1 ALBEDO, 2 PBR3, 3 ALL
static inline uint32_t r3n_material_features(const r3n_material208 &m) {
    const uint32_t f = m.flags;
    uint32_t k = 0;
    if (m.textures[0]) k |= R3N_FEAT_TEX_ALBEDO;
    if (m.textures[1]) k |= R3N_FEAT_TEX_NORMAL;
    if (m.textures[2]) k |= R3N_FEAT_TEX_AOMR;
    if (m.textures[3] || m.textures[9]) k |= R3N_FEAT_TEX_AOMR_X;
    if (m.textures[4] || m.textures[5] || m.textures[6] || m.textures[7]) k |= R3N_FEAT_TEX_MISC;
    if (f & R3N_FLAGS_ALBEDO_BLEND) k |= R3N_FEAT_VCOLOR;
    if (f & R3N_FLAGS_UNLIT) k |= R3N_FEAT_UNLIT;
    if (f & R3N_FLAGS_NEAREST) k |= R3N_FEAT_NEAREST;
    if ((m.textures[2] || m.textures[3] || m.textures[9]) && !(f & R3N_FLAGS_AOMR_COMBINED)) k |= R3N_FEAT_AOMR_SPLIT;
    if (!(m.clear_coat == 0.0f)) k |= R3N_FEAT_CLEARCOAT;  // (NaN counts)
    if (f & (R3N_FLAGS_BICOMPONENT_NORMAL | R3N_FLAGS_SWIZZLED_NORMAL | R3N_FLAGS_YDOWN_NORMAL)) k |= R3N_FEAT_NORMAL_FLAGS;
    if (!(f & R3N_FLAGS_ALBEDO_ACTIVE)) k |= R3N_FEAT_ALBEDO_OFF;
    return k;
}
// smallest variant of the chain whose feature set covers `feat`
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline uint32_t r3n_variant_of(uint32_t feat) {
    if (feat & ~R3N_CLS_PBR3) return 3u;
    if (feat & ~R3N_CLS_ALBEDO) return 2u;
    return feat ? 1u : 0u;
}

// What the vertex stage (opaque.wgsl:91-135) and the triangle setup produce for one triangle (see vertex_stage below).
struct TriRecord {
    float e[3][3];      // oriented edge functions of the triangle setup
    float vp[3][4];     // view-space positions
    float vn[3][3];     // view-space normals (normalised per vertex)
    float vt[3][3];     // view-space tangents (only when the material has a normal map, else 0)
    float vc[3][4];     // vertex colours
    float uv[3][2];     // texture coordinates 0
    uint32_t object, material;
    uint32_t feat;      // R3N_FEAT_* of the material (ShadeArgs::material_feat), R3N_FEAT_ALL without the table
    uint32_t _pad[4];
};
static_assert(sizeof(TriRecord) == 256, "triangle record is 64 dwords");

// Transparent pass, stage 3 (see k_blend_apply below).
struct BlendApplyArgs {
    const unsigned long long *keys;   // nodes: next << 32 | draw order
    const uint32_t *vals;             // nodes: canonical slot + 1
    const uint32_t *head;             // per pixel sample: first node, R3N_INVALID = none
    uint32_t first_sample, n_samples; // the samples of the rows this rank resolves
    uint32_t capacity;                // nodes the pass can hold: no list is longer, no sample is blended more often (loop bounds)
    ushort4 *samples;  // S == 1: the HDR target itself; S == 4: the per-sample colours
};

#define R3N_SRGB_LUT_SIZE 0x3C00u  // half bits of 1.0

// The light lists in view space, as the fragment stage reads them (stage_lights below).
struct LdsDirLight {
    float m[16];      // light.view_proj * uniforms.inv_view (opaque.wgsl:491)
    float l[3];       // normalize(view_mat3 * -direction)   (opaque.wgsl:519)
    float color[3];
    float inv_res[2], offset[2], size[2];
    float sane;       // 1: |colour| <= 1e6 (lets the fragment stage skip fully occluded lights), else 0
    float _pad[3];
};
static_assert(sizeof(LdsDirLight) == 128, "two s_load_dwordx16");
struct LdsPointLight {
    float vpos[3];    // (uniforms.view * position).xyz (opaque.wgsl:528)
    float color[3];
    float radius;
    float _pad;
};
static_assert(sizeof(LdsPointLight) == 32, "one s_load_dwordx8");
// The same lists in GLOBAL memory, written once per frame by k_stage_view_lights: the single-sample resolve then reads a light
// through scalar loads -- the light index is wave-uniform -- into scalar registers: no per-workgroup staging pass and barrier in
// front of 32 400 workgroups, no LDS reads and no thirty vector registers per light in the light loop.
struct ViewLights {
    uint32_t n_dir, n_point, _pad[2];
    LdsDirLight dir[R3N_MAX_DIR_LIGHTS];
    LdsPointLight point[R3N_MAX_POINT_LIGHTS];
};

#ifndef R3N_SHADE_DECL_ONLY

// shadow/pcf.wgsl + comparison sampler (samplers.rs:24,42-57): bilinear, GreaterEqual, Repeat.
// Texel coordinates + bilinear weights of one comparison tap, exactly as sample_compare derives them.
struct PcfTap {
    int ix, iy;  // |floor| < 1e9 fits
    float fx, fy;
};
R3N_DEV PcfTap pcf_tap(uint32_t aw, uint32_t ah, float u, float v, int ox, int oy) {
    const float tx = (u * (float)aw - 0.5f) + (float)ox;
    const float ty = (v * (float)ah - 0.5f) + (float)oy;
    const float fx0 = floorf(tx), fy0 = floorf(ty);
    PcfTap t;
    t.fx = tx - fx0; t.fy = ty - fy0;
    t.ix = (fx0 == fx0 && fabsf(fx0) < 1e9f) ? (int)fx0 : 0;
    t.iy = (fy0 == fy0 && fabsf(fy0) < 1e9f) ? (int)fy0 : 0;
    if (!(t.fx == t.fx)) t.fx = 0.0f;
    if (!(t.fy == t.fy)) t.fy = 0.0f;
    return t;
}
// Repeat addressing (samplers.rs:24): the texel index is almost always already inside the atlas
R3N_DEV uint32_t wrap_texel(int v, uint32_t n) {
    if ((uint32_t)v < n) return (uint32_t)v;
    const long long w = (long long)n;
    return (uint32_t)((((long long)v % w) + w) % w);
}
R3N_DEV float pcf_texel_cmp(const float *__restrict__ atlas, uint32_t aw, uint32_t ah, int x, int y, float ref) {
    return ref >= atlas[(size_t)wrap_texel(y, ah) * aw + wrap_texel(x, aw)] ? 1.0f : 0.0f;
}

// shadow/pcf.wgsl: mean of 5 bilinear comparison taps (centre, +-1 texel in x and y).  The 5 taps touch 20 texels
// of which only 12 are distinct (a 4x4 block without its corners): the comparisons are fetched once and every tap
// then applies its own weights -- same values, same operation order as five independent sample_compare calls.
// This is the path nobody takes (the regular form below covers every lookup whose 4x4 block lies inside the atlas), so it is
// written for SIZE: rolled loops, one texel fetch site.  Unrolled, with a 64-bit Repeat remainder per texel, it was 13 000 of the
// light loop's 13 800 instructions and its live ranges weighed on the register allocation of the whole fragment stage.
template <class M>
R3N_DEV float shadow_pcf5_general(const float *__restrict__ atlas, uint32_t aw, uint32_t ah, float u, float v, float ref) {
    float r = 0.0f;
#pragma unroll 1
    for (int k = 0; k < 5; ++k) {
        // tap offsets (0, 0), (0, 1), (0, -1), (1, 0), (-1, 0)
        const int ox = k == 3 ? 1 : (k == 4 ? -1 : 0), oy = k == 1 ? 1 : (k == 2 ? -1 : 0);
        const PcfTap t = pcf_tap(aw, ah, u, v, ox, oy);
        float c[4];
#pragma unroll 1
        for (int j = 0; j < 4; ++j) c[j] = pcf_texel_cmp(atlas, aw, ah, t.ix + (j & 1), t.iy + (j >> 1), ref);
        // top = c00 * (1 - fx) + c10 * fx, bot = c01 * (1 - fx) + c11 * fx as one packed pair
        const f2 tb = M::mad((f2){c[1], c[3]}, splat2(t.fx), (f2){c[0], c[2]} * splat2(1.0f - t.fx));
        r = r + M::mad(tb.y, t.fy, tb.x * (1.0f - t.fy));
    }
    return r * 0.2f;
}

// The same five taps when nothing unusual happens -- every floor is a small integer, the +-1 texel taps land one texel
// away, the 4x4 block lies inside the atlas: ONE test for the whole lookup instead of one per tap and per texel, 32-bit offsets
// from the uniform atlas pointer, no Repeat wrap.  Same comparisons, weights and summation order as the general form.
template <class M>
R3N_DEV float shadow_pcf5(const float *__restrict__ atlas, uint32_t aw, uint32_t ah, float u, float v, float ref) {
    const float tx0 = u * (float)aw - 0.5f, ty0 = v * (float)ah - 0.5f;
    const float txp = tx0 + 1.0f, txm = tx0 + -1.0f, typ = ty0 + 1.0f, tym = ty0 + -1.0f;
    const float f0x = floorf(tx0), fpx = floorf(txp), fmx = floorf(txm);
    const float f0y = floorf(ty0), fpy = floorf(typ), fmy = floorf(tym);
    const bool regular = fpx == f0x + 1.0f && fmx == f0x - 1.0f && fpy == f0y + 1.0f && fmy == f0y - 1.0f &&
                         f0x >= 1.0f && f0x <= (float)aw - 3.0f && f0y >= 1.0f && f0y <= (float)ah - 3.0f &&
                         ((unsigned long long)aw * ah <= (1ull << 30));  // (NaN fails every comparison)
    if (!regular) return shadow_pcf5_general<M>(atlas, aw, ah, u, v, ref);
    const float fx0 = tx0 - f0x, fxp = txp - fpx, fxm = txm - fmx;
    const float fy0 = ty0 - f0y, fyp = typ - fpy, fym = tym - fmy;
    // cmp[dy][dx] for texel (cx - 1 + dx, cy - 1 + dy); the corners are never read
    const uint32_t row = aw << 2;
    const uint32_t base = (__umul24((uint32_t)(int)f0y - 1u, aw) + ((uint32_t)(int)f0x - 1u)) << 2;  // byte offset of texel (cx - 1, cy - 1)
    const char *ap = reinterpret_cast<const char *>(atlas);
    // the 12 texels as FOUR loads (2 + 4 + 4 + 2 texels: the block's rows are contiguous; dword-aligned multi-dword loads) instead of
    // twelve: a third of the vector-memory instructions of the whole fragment stage were these
    struct __attribute__((packed, aligned(4))) T2 { float v[2]; };
    struct __attribute__((packed, aligned(4))) T4 { float v[4]; };
    const T2 r0 = *reinterpret_cast<const T2 *>(ap + (base + 4u));
    const T4 r1 = *reinterpret_cast<const T4 *>(ap + (base + row));
    const T4 r2 = *reinterpret_cast<const T4 *>(ap + (base + 2u * row));
    const T2 r3 = *reinterpret_cast<const T2 *>(ap + (base + 3u * row + 4u));
    // The comparison results stay BOOLEANS (lane masks in scalar registers): a tap's c * weight with c in {0, 1} and a finite
    // weight >= 0 is the weight or +0 exactly, i.e. a select -- no 1.0 / 0.0 floats, no multiplies by them.
    const bool c01 = ref >= r0.v[0], c02 = ref >= r0.v[1];
    const bool c10 = ref >= r1.v[0], c11 = ref >= r1.v[1], c12 = ref >= r1.v[2], c13 = ref >= r1.v[3];
    const bool c20 = ref >= r2.v[0], c21 = ref >= r2.v[1], c22 = ref >= r2.v[2], c23 = ref >= r2.v[3];
    const bool c31 = ref >= r3.v[0], c32 = ref >= r3.v[1];
    // tap = (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy with the products as selects (fx, fy in [0, 1) on this path)
    auto tapb = [&](bool a, bool b, bool c, bool d, float fx, float fy) {
        const float omx = 1.0f - fx;
        const f2 tb = (f2){b ? fx : 0.0f, d ? fx : 0.0f} + (f2){a ? omx : 0.0f, c ? omx : 0.0f};
        return M::mad(tb.y, fy, tb.x * (1.0f - fy));
    };
    {
        float r = 0.0f;
        r = r + tapb(c11, c12, c21, c22, fx0, fy0);  // ( 0,  0)
        r = r + tapb(c21, c22, c31, c32, fx0, fyp);  // ( 0, +1)
        r = r + tapb(c01, c02, c11, c12, fx0, fym);  // ( 0, -1)
        r = r + tapb(c12, c13, c22, c23, fxp, fy0);  // (+1,  0)
        r = r + tapb(c10, c11, c20, c21, fxm, fy0);  // (-1,  0)
        return r * 0.2f;
    }
}

struct PixelData {
    float albedo[4], diffuse[3], roughness, normal[3], f0[3], emissive[3], ao;
};

#define R3N_PI 3.14159265359f

// The terms of surface_shading that do not depend on the light, evaluated once per pixel instead of once per light (same
// expressions, same values: a square root, two dot products and a handful of multiplies per further light).
struct BrdfPixel {
    float nov, a2, f90, sqrt_v;  // N.V + 1e-5, roughness^2, the Fresnel f90, sqrt((-nov * a2 + nov) * nov + a2) of the visibility term
};
template <class M>
R3N_DEV BrdfPixel brdf_pixel(const PixelData &px, const float v[3]) {
    BrdfPixel b;
    b.nov = fabsf(dot3m<M>(px.normal, v)) + 0.00001f;
    const float c165[3] = {16.5f, 16.5f, 16.5f};
    b.f90 = sat(dot3m<M>(px.f0, c165));
    const float a = px.roughness;
    b.a2 = a * a;
    b.sqrt_v = M::sqrt(M::mad(M::mad(-b.nov, b.a2, b.nov), b.nov, b.a2));
    return b;
}
// opaque.wgsl:440-468.  The r and g channels run as one packed pair, b on its own.
template <class M>
R3N_DEV void surface_shading(const float l[3], const float intensity[3], const PixelData &px, const float v[3],
                             float occlusion, float out[3], const BrdfPixel *pre = nullptr) {
    float h[3] = {v[0] + l[0], v[1] + l[1], v[2] + l[2]};
    normalize3m<M>(h);
    const float nov = pre ? pre->nov : fabsf(dot3m<M>(px.normal, v)) + 0.00001f;
    const float nol = sat(dot3m<M>(px.normal, l));
    const float noh = sat(dot3m<M>(px.normal, h));
    const float loh = sat(dot3m<M>(l, h));
    const float c165[3] = {16.5f, 16.5f, 16.5f};
    const float f90 = pre ? pre->f90 : sat(dot3m<M>(px.f0, c165));
    const float a = px.roughness, a2 = pre ? pre->a2 : a * a;
    const float f = M::mad(M::mad(noh, a2, -noh), noh, 1.0f);             // (noh * a2 - noh) * noh + 1
    const float d = M::div(a2, (R3N_PI * f) * f);
    const float x = 1.0f - loh, x2 = x * x, x5 = (x2 * x2) * x;
    const float ggxl = nov * M::sqrt(M::mad(M::mad(-nol, a2, nol), nol, a2));  // (-nol * a2 + nol) * nol + a2
    const float ggxv = nol * (pre ? pre->sqrt_v : M::sqrt(M::mad(M::mad(-nov, a2, nov), nov, a2)));
    const float vis = M::half_over(ggxl + ggxv);  // 0.5 / (ggxl + ggxv)
    const float k = nol * occlusion;
    const float dv = d * vis;
    {
        const f2 f0 = {px.f0[0], px.f0[1]};
        const f2 fres = M::mad(splat2(f90) - f0, splat2(x5), f0);          // f0 + (f90 - f0) * x5
        const f2 color = M::mad(splat2(dv), fres, (f2){px.diffuse[0], px.diffuse[1]} * splat2(1.0f / R3N_PI));  // fd + fr
        const f2 o = (color * (f2){intensity[0], intensity[1]}) * splat2(k);
        out[0] = o.x; out[1] = o.y;
    }
    {
        const float fres = M::mad(f90 - px.f0[2], x5, px.f0[2]);
        const float color = M::mad(dv, fres, px.diffuse[2] * (1.0f / R3N_PI));
        out[2] = (color * intensity[2]) * k;
    }
}

R3N_DEV float srgb_to_linear(float e) { return e > 0.04045f ? powf((e + 0.055f) / 1.055f, 2.4f) : e / 12.92f; }

R3N_DEV float srgb_oetf(float x) {
    if (!(x > 0.0f)) return 0.0f;
    if (x >= 1.0f) return 1.0f;
    if (x <= 0.0031308f) return x * 12.92f;
    return 1.055f * powf(x, 1.0f / 2.4f) - 0.055f;
}
// The HDR target is Rgba16Float, so the OETF input is one of 65536 half values and only those in (0, 1) need the
// formula: 15360 bit patterns.  k_build_srgb_lut evaluates the exact expression once per pattern at context
// creation; the per-pixel path is then a byte gather instead of three powf (measured: 73 us of the 4K resolve).
static __global__ __launch_bounds__(256) void k_build_srgb_lut(unsigned char *__restrict__ lut) {
    const uint32_t h = blockIdx.x * 256u + threadIdx.x;
    if (h >= R3N_SRGB_LUT_SIZE) return;
    const float x = (float)__builtin_bit_cast(_Float16, (unsigned short)h);
    lut[h] = (unsigned char)(srgb_oetf(x) * 255.0f + 0.5f);
}
// (unsigned char)(srgb_oetf(x) * 255 + 0.5) for the half with bit pattern h
R3N_DEV unsigned char srgb8_of_half(const unsigned char *__restrict__ lut, unsigned short h) {
    if (h & 0x8000u) return 0;       // negative, -0, negative NaN: !(x > 0)
    if (h > 0x7C00u) return 0;       // NaN
    if (h >= R3N_SRGB_LUT_SIZE) return 255;  // x >= 1 (and +inf)
    return lut[h];
}
// blit.wgsl fs_main_scene into an Rgba8UnormSrgb target: exact OETF of the Rgba16Float-rounded value
// bgr: the target is a Bgra8* format (blue first in memory)
R3N_DEV uchar4 tonemap_half4(const unsigned char *__restrict__ lut, ushort4 h, bool bgr = false) {
    const float al = (float)__builtin_bit_cast(_Float16, h.w);
    const float a = (!(al > 0.0f)) ? 0.0f : (al >= 1.0f ? 1.0f : al);
    const unsigned char r = srgb8_of_half(lut, h.x), g = srgb8_of_half(lut, h.y), b = srgb8_of_half(lut, h.z);
    return make_uchar4(bgr ? b : r, g, bgr ? r : b, (unsigned char)(a * 255.0f + 0.5f));
}
// math/color.wgsl:13-19 srgb_scene_to_display + the unorm store's clamp: what blit.wgsl fs_main_monitor writes into a
// target whose format is not *Srgb (tonemapping.rs:44)
R3N_DEV float srgb_scene_to_display(float x) {
    const float e = x > 0.0031308f ? 1.055f * powf(x, 0.4166f) - 0.055f : x * 12.92f;
    if (!(e > 0.0f)) return 0.0f;
    return e >= 1.0f ? 1.0f : e;
}

R3N_DEV ushort4 pack_half4(const float v[4]) {
    ushort4 o;
    // float -> half conversion rounds to nearest even (v_cvt_f16_f32)
    o.x = __builtin_bit_cast(unsigned short, (_Float16)v[0]);
    o.y = __builtin_bit_cast(unsigned short, (_Float16)v[1]);
    o.z = __builtin_bit_cast(unsigned short, (_Float16)v[2]);
    o.w = __builtin_bit_cast(unsigned short, (_Float16)v[3]);
    return o;
}

// opaque.wgsl VS (:91-135) + FS (:203-551) for triangle slot `id - 1` at the centre of pixel (x, y).
// What the vertex stage (opaque.wgsl:91-135) and the triangle setup produce for one triangle: everything the fragment
// stage needs that does not depend on the pixel.  64 floats = 256 B.  With one sample per pixel the resolve does not
// recompute this per pixel: k_mark_visible flags the triangles that own a pixel, k_vertex_stage evaluates the record once
// per flagged triangle, the per-pixel kernel loads it (neighbouring pixels share it).  Same arithmetic either way.

template <bool TEX>
R3N_DEV void vertex_stage(const ShadeArgs &a, uint32_t id, TriRecord &r) {
    const uint32_t slot = id - 1u;
    // object = last o with tri_base[o] <= slot; the coarse table narrows the binary search to the objects that
    // start inside one 256-slot bucket (usually zero or one step instead of log2(capacity))
    const uint32_t bucket = slot >> R3N_SLOT_TABLE_SHIFT;
    uint32_t lo = a.slot_table[bucket];
    uint32_t hi = bucket + 1u < a.slot_table_size ? a.slot_table[bucket + 1u] + 1u : a.hdr->object_count;
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (a.tri_base[mid] <= slot) lo = mid; else hi = mid;
    }
    const uint32_t obj = lo, tri = slot - a.tri_base[obj];
    const r3n_object128 &ob = a.objects[obj];
    // The record's index / material / attribute-offset fields (bytes 80..115) in three loads issued together, and the material's ten
    // texture ids the same way below: read field by field -- the offsets one per attribute, the ids through a short-circuit `||` --
    // they were up to a dozen DEPENDENT round trips in front of the vertex fetches of a kernel that is nothing but its load chain.
    const uint4 of0 = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(&ob) + 80);   // first_index, index_count, material_index, offsets[0]
    const uint4 of1 = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(&ob) + 96);   // offsets[1..4]
    const uint32_t of5 = ob.vertex_attribute_start_offsets[5];
    const uint32_t mat_index = of0.z < a.n_materials ? of0.z : 0u;
    const r3n_material208 &mat = a.materials[mat_index];
    const float *mv = a.baked[obj].model_view;
    r.object = obj;
    r.material = mat_index;
    r.feat = a.material_feat != nullptr ? a.material_feat[mat_index] : R3N_FEAT_ALL;

    // vertex stage for the 3 vertices (opaque.wgsl:114-134)
    uint32_t idx[3];
    float p[3][4];
    const float inv_s2[3] = {1.0f / dot3(mv, mv), 1.0f / dot3(mv + 4, mv + 4), 1.0f / dot3(mv + 8, mv + 8)};
    const uint32_t first = of0.x + tri * 3u;
    const uint32_t pos_off = of0.w;
    const uint32_t nrm_off = of1.x;
    const uint32_t tan_off = of1.y;
    const uint32_t uv0_off = of1.z;
    const uint32_t col_off = of5;
    bool any_tex = false, normal_map = false;
    if (TEX) {
        const uint4 t0 = *reinterpret_cast<const uint4 *>(&mat.textures[0]), t1 = *reinterpret_cast<const uint4 *>(&mat.textures[4]);
        const uint2 t2 = *reinterpret_cast<const uint2 *>(&mat.textures[8]);
        any_tex = (((t0.x | t0.y) | (t0.z | t0.w)) | ((t1.x | t1.y) | (t1.z | t1.w)) | (t2.x | t2.y)) != 0u;
        normal_map = t0.y != 0u;
    }
    fetch_indices3(a.mesh, first, idx);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v[3];
        fetch_vec3(a.mesh, pos_off, idx[k], v);
        mul_point(a.baked[obj].model_view_proj, v, p[k]);
        mul_point(mv, v, r.vp[k]);
        float nm[3] = {0.0f, 0.0f, 0.0f};
        if (nrm_off != R3N_INVALID) fetch_vec3(a.mesh, nrm_off, idx[k], nm);
        const float sn[3] = {inv_s2[0] * nm[0], inv_s2[1] * nm[1], inv_s2[2] * nm[2]};
        mat3_mul_vec3(mv, mv + 4, mv + 8, sn, r.vn[k]);
        normalize3(r.vn[k]);
        if (TEX && normal_map) {  // vs_out.tangent (opaque.wgsl:129); only the normal map reads it
            float tg[3] = {0.0f, 0.0f, 0.0f};
            if (tan_off != R3N_INVALID) fetch_vec3(a.mesh, tan_off, idx[k], tg);
            const float st[3] = {inv_s2[0] * tg[0], inv_s2[1] * tg[1], inv_s2[2] * tg[2]};
            mat3_mul_vec3(mv, mv + 4, mv + 8, st, r.vt[k]);
            normalize3(r.vt[k]);
        } else {
            r.vt[k][0] = r.vt[k][1] = r.vt[k][2] = 0.0f;
        }
        if (col_off != R3N_INVALID) {
            const uint32_t cw = a.mesh[col_off / 4u + idx[k]];
#pragma unroll
            for (int c = 0; c < 4; ++c) r.vc[k][c] = (float)((cw >> (8 * c)) & 0xFFu) / 255.0f;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) r.vc[k][c] = 1.0f;
        }
        if (TEX && any_tex) fetch_uv0(a.mesh, uv0_off, idx[k], r.uv[k]);
        else r.uv[k][0] = r.uv[k][1] = 0.0f;
    }
    TriSetup ts;
    setup_triangle(p, (float)a.width / 2.0f, (float)a.height / 2.0f,
                   (a.hdr->flags & R3N_PCU_POSITIVE_AREA_VISIBLE) != 0u, ts);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) r.e[i][c] = ts.e[i][c];
}

// (lam0 * a0 + lam1 * a1) + lam2 * a2 for a pair of attribute channels
template <class M> R3N_DEV f2 interp2(const float lam[3], f2 a0, f2 a1, f2 a2) {
    return M::mad(splat2(lam[2]), a2, M::mad(splat2(lam[1]), a1, splat2(lam[0]) * a0));
}
template <class M> R3N_DEV float interp1(const float lam[3], float a0, float a1, float a2) {
    return M::mad(lam[2], a2, M::mad(lam[1], a1, lam[0] * a0));
}
template <class M> R3N_DEV void interp_vec3(const float lam[3], const float a[3][3], float o[3]) {
    const f2 xy = interp2<M>(lam, (f2){a[0][0], a[0][1]}, (f2){a[1][0], a[1][1]}, (f2){a[2][0], a[2][1]});
    o[0] = xy.x; o[1] = xy.y;
    o[2] = interp1<M>(lam, a[0][2], a[1][2], a[2][2]);
}
template <class M> R3N_DEV void interp_vec4(const float lam[3], const float a[3][4], float o[4]) {
    const f2 xy = interp2<M>(lam, (f2){a[0][0], a[0][1]}, (f2){a[1][0], a[1][1]}, (f2){a[2][0], a[2][1]});
    const f2 zw = interp2<M>(lam, (f2){a[0][2], a[0][3]}, (f2){a[1][2], a[1][3]}, (f2){a[2][2], a[2][3]});
    o[0] = xy.x; o[1] = xy.y; o[2] = zw.x; o[3] = zw.y;
}

// opaque.wgsl FS (:203-551) for the triangle record `r` at the centre of pixel (x, y).
// M = MathExact: the arithmetic contract, bit-identical to the oracle.  M = MathFast (opt-in): fused multiply-adds and
// hardware reciprocal / rsqrt in the interpolation, filtering and BRDF arithmetic; the edge functions, the level-of-detail /
// footprint selection, the shadow coordinates and the depth comparisons stay exact so that no pixel changes triangle, mip
// level or shadow texel -- the result differs from the exact one by rounding only.
// CLS: the features (R3N_FEAT_*) a material shaded here may have; the code of every other feature is not instantiated.
// SL: s_dir / s_point are in global memory and a light is fetched with scalar loads (ViewLights); else they are the workgroup's LDS copies.
template <class T, bool SL> R3N_DEV T load_light(const T *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (SL) {  // wave-uniform address + constant address space: s_load into scalar registers
        static_assert(sizeof(T) % 4 == 0, "whole dwords");
        const unsigned long long v = (unsigned long long)p;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        typedef __attribute__((address_space(4))) const uint32_t *sp_t;
        const sp_t sp = (sp_t)(((unsigned long long)hi << 32) | (unsigned long long)lo);
        union { T t; uint32_t w[sizeof(T) / 4]; } u;
#pragma unroll
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) u.w[k] = sp[k];
        return u.t;
    }
#endif
    return *p;
}
template <bool TEX, class M = MathExact, uint32_t CLS = R3N_CLS_ALL, bool SL = false>
R3N_DEV void fragment_stage(const ShadeArgs &a, const LdsDirLight *s_dir, const LdsPointLight *s_point, uint32_t n_dir,
                            uint32_t n_point, const TriRecord &r, uint32_t x, uint32_t y, float out[4]) {
    const r3n_material208 &mat = a.materials[r.material];
    constexpr bool kVColor = (CLS & R3N_FEAT_VCOLOR) != 0u, kUnlit = (CLS & R3N_FEAT_UNLIT) != 0u, kNearest = (CLS & R3N_FEAT_NEAREST) != 0u;
    constexpr bool kSplit = (CLS & R3N_FEAT_AOMR_SPLIT) != 0u, kNormalFlags = (CLS & R3N_FEAT_NORMAL_FLAGS) != 0u;
    constexpr bool kAlbedoOff = (CLS & R3N_FEAT_ALBEDO_OFF) != 0u, kMisc = (CLS & R3N_FEAT_TEX_MISC) != 0u;
    constexpr bool kClearCoat = (CLS & (R3N_FEAT_CLEARCOAT | R3N_FEAT_TEX_MISC)) != 0u;
    constexpr bool kAnyTex = TEX && (CLS & (R3N_FEAT_TEX_ALBEDO | R3N_FEAT_TEX_NORMAL | R3N_FEAT_TEX_AOMR | R3N_FEAT_TEX_AOMR_X | R3N_FEAT_TEX_MISC)) != 0u;
    // texture slot -> the feature that binds it
    constexpr uint32_t kSlotFeat[10] = {R3N_FEAT_TEX_ALBEDO, R3N_FEAT_TEX_NORMAL, R3N_FEAT_TEX_AOMR, R3N_FEAT_TEX_AOMR_X, R3N_FEAT_TEX_MISC,
                                        R3N_FEAT_TEX_MISC, R3N_FEAT_TEX_MISC, R3N_FEAT_TEX_MISC, R3N_FEAT_TEX_MISC /* anisotropy: unread */,
                                        R3N_FEAT_TEX_AOMR_X};
    TriSetup ts;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) ts.e[i][c] = r.e[i][c];
    ts.z[0] = ts.z[1] = ts.z[2] = 0.0f; ts.det = 1.0f; ts.valid = true;  // not used by the fragment stage
    float E[3];
    (void)edge_eval(ts, (float)x + 0.5f, (float)y + 0.5f, E);
    // barycentrics and the view-space position: exact arithmetic under both policies.  The position feeds the shadow
    // coordinates and the comparison depth, and without a depth bias (reference behaviour) a lit surface compares against its
    // own rasterised depth -- a last-bit change of the position flips comparisons all over it.
    const float rs = exact_math::rcp((E[0] + E[1]) + E[2]);  // 1 / sum, correctly rounded under both policies
    const float lam[3] = {E[0] * rs, E[1] * rs, E[2] * rs};
    float vpos[4], nrm[3], col[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    interp_vec4<MathExact>(lam, r.vp, vpos);
    interp_vec3<M>(lam, r.vn, nrm);
    if (kVColor && (mat.flags & R3N_FLAGS_ALBEDO_ACTIVE) && (mat.flags & R3N_FLAGS_ALBEDO_BLEND))  // the only reader of vs_out.color
        interp_vec4<M>(lam, r.vc, col);

    // fragment stage (opaque.wgsl:203-424).  Texture slots (managers/material.rs:25-29 order): 0 albedo, 1 normal,
    // 2 roughness, 3 metallic, 4 reflectance, 5 clear coat, 6 clear coat roughness, 7 emissive, 8 anisotropy, 9 AO
    PixelData px;
    const uint32_t mflags = mat.flags;
    bool any_tex = false;
    if (kAnyTex) {
#pragma unroll
        for (int k = 0; k < 10; ++k)
            if (CLS & kSlotFeat[k]) any_tex = any_tex || mat.textures[k] != 0u;
    }
    float coords[2] = {0.0f, 0.0f}, ddx[2] = {0.0f, 0.0f}, ddy[2] = {0.0f, 0.0f};
    const bool nearest = kNearest && (mflags & R3N_FLAGS_NEAREST) != 0u;
    if (kAnyTex && any_tex) {  // opaque.wgsl:207-209
        const f2 sr = interp2<M>(lam, (f2){r.uv[0][0], r.uv[0][1]}, (f2){r.uv[1][0], r.uv[1][1]}, (f2){r.uv[2][0], r.uv[2][1]});
        const float self_raw[2] = {sr.x, sr.y};
        frag_coords<M>(ts, r.uv, mat.uv_transform0, (int)x, (int)y, coords, ddx, ddy, self_raw);
    }
    // tex4: all four channels; tex3: the callers that read r, g, b only (alpha neither decoded nor filtered)
    // (a class without TEX_GENERAL / NEAREST: every bound texture is on the sampler's short path -- the host's census says so)
    constexpr bool kShortOnly = (CLS & (R3N_FEAT_TEX_GENERAL | R3N_FEAT_NEAREST)) == 0u;
    // classes that can bind more than one map: the maps share level of detail and footprints where their extents allow (texture.h TexShare)
    constexpr bool kShare = kShortOnly && (CLS & (R3N_FEAT_TEX_NORMAL | R3N_FEAT_TEX_AOMR)) != 0u;
    TexShare tshare;
    tshare.width = 0u; tshare.height = 0u; tshare.mips = 0u; tshare.level = 0u; tshare.frac = 0.0f;
    auto tex4 = [&](int slot, float dst[4]) { tex_sample_grad<M, true, kShortOnly, kShare>(a.tex, mat.textures[slot], nearest, coords[0], coords[1], ddx, ddy, dst, &tshare); };
    auto tex3 = [&](int slot, float dst[4]) { tex_sample_grad<M, false, kShortOnly, kShare>(a.tex, mat.textures[slot], nearest, coords[0], coords[1], ddx, ddy, dst, &tshare); };
    auto has = [&](int slot) { return TEX && (CLS & kSlotFeat[slot]) != 0u && mat.textures[slot] != 0u; };
    // the PBR class (base colour + normal + AO / roughness / metallic maps, nothing else): its three maps in one batched pass
    constexpr bool kBatch = TEX && kShortOnly && CLS == R3N_CLS_PBR3;
    float tb3[3][4];
    bool batched = false;
    if (kBatch && any_tex) {
        const uint32_t ids[3] = {mat.textures[0], mat.textures[1], mat.textures[2]};
        batched = tex_sample3_batched<M>(a.tex, ids, coords[0], coords[1], ddx, ddy, tb3);  // (wave-uniform)
    }
    auto tex_slot012 = [&](int slot, float dst[4], bool need_a) {
        if (kBatch && batched) {
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[c] = tb3[slot][c];
        } else if (need_a) tex4(slot, dst);
        else tex3(slot, dst);
    };
    if (!kAlbedoOff || (mflags & R3N_FLAGS_ALBEDO_ACTIVE)) {
#pragma unroll
        for (int c = 0; c < 4; ++c) px.albedo[c] = 1.0f;
        if (has(0)) tex_slot012(0, px.albedo, true);
        if (kVColor && (mflags & R3N_FLAGS_ALBEDO_BLEND)) {
            if (mflags & R3N_FLAGS_ALBEDO_VERTEX_SRGB) {
#pragma unroll
                for (int c = 0; c < 3; ++c) px.albedo[c] *= srgb_to_linear(col[c]);
                px.albedo[3] *= col[3];
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) px.albedo[c] *= col[c];
            }
        }
    } else {
        px.albedo[0] = px.albedo[1] = px.albedo[2] = 0.0f;
        px.albedo[3] = 1.0f;
    }
    {
        const f2 rg = (f2){px.albedo[0], px.albedo[1]} * (f2){mat.albedo[0], mat.albedo[1]};
        const f2 ba = (f2){px.albedo[2], px.albedo[3]} * (f2){mat.albedo[2], mat.albedo[3]};
        px.albedo[0] = rg.x; px.albedo[1] = rg.y; px.albedo[2] = ba.x; px.albedo[3] = ba.y;
    }

    if (kUnlit && (mflags & R3N_FLAGS_UNLIT)) {
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = px.albedo[c];
    } else {
        // --- normal (opaque.wgsl:246-273)
        if (has(1)) {
            float t[4], n[3];
            tex_slot012(1, t, true);
            if (kNormalFlags && (mflags & R3N_FLAGS_BICOMPONENT_NORMAL)) {
                float b0 = (mflags & R3N_FLAGS_SWIZZLED_NORMAL) ? t[3] : t[0], b1 = t[1];  // texture_read.ag : .rg
                b0 = M::mad(b0, 2.0f, -1.0f);
                b1 = M::mad(b1, 2.0f, -1.0f);
                n[0] = b0; n[1] = b1;
                n[2] = M::sqrt((1.0f - b0 * b0) - b1 * b1);
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) n[c] = M::mad(t[c], 2.0f, -1.0f);
                normalize3m<M>(n);
            }
            if (kNormalFlags && (mflags & R3N_FLAGS_YDOWN_NORMAL)) n[1] = -n[1];
            float tng[3];
            interp_vec3<M>(lam, r.vt, tng);
            float nn[3] = {nrm[0], nrm[1], nrm[2]};
            normalize3m<M>(nn);
            normalize3m<M>(tng);
            const float bt[3] = {nn[1] * tng[2] - tng[1] * nn[2], nn[2] * tng[0] - tng[2] * nn[0], nn[0] * tng[1] - tng[0] * nn[1]};
            mat3_mul_vec3m<M>(tng, bt, nn, n, px.normal);  // tbn * normal
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) px.normal[c] = nrm[c];
        }
        normalize3m<M>(px.normal);
        // --- AO, metallic, roughness (opaque.wgsl:277-351)
        float ao = mat.ambient_occlusion, pr = mat.roughness, metallic = mat.metallic;
        if (!kSplit || (mflags & R3N_FLAGS_AOMR_COMBINED)) {  // (!kSplit: a bound slot 2 implies the combined layout)
            if (has(2)) {
                float t[4];
                tex_slot012(2, t, false);
                ao = mat.ambient_occlusion * t[0];
                pr = mat.roughness * t[1];
                metallic = mat.metallic * t[2];
            }
        } else if (mflags & R3N_FLAGS_AOMR_BW_SPLIT) {
            float t[4];
            if (has(2)) { tex3(2, t); pr = mat.roughness * t[0]; }
            if (has(3)) { tex3(3, t); metallic = mat.metallic * t[0]; }
            if (has(9)) { tex3(9, t); ao = mat.ambient_occlusion * t[0]; }
        } else {
            float t[4];
            if (has(2)) {
                tex3(2, t);
                const bool sw = (mflags & R3N_FLAGS_AOMR_SWIZZLED_SPLIT) != 0u;
                pr = mat.roughness * (sw ? t[1] : t[0]);
                metallic = mat.metallic * (sw ? t[2] : t[1]);
            }
            if (has(9)) { tex3(9, t); ao = mat.ambient_occlusion * t[0]; }
        }
        // --- reflectance (opaque.wgsl:355-359)
        float reflectance = mat.reflectance;
        if (has(4)) { float t[4]; tex3(4, t); reflectance = mat.reflectance * t[0]; }
        // --- clear coat (opaque.wgsl:363-391)
        float cc = mat.clear_coat, ccpr = mat.clear_coat_roughness;
        if (!kMisc) {
            // slots 5 / 6 unbound: the factors as they are
        } else if (mflags & R3N_FLAGS_CC_GLTF_COMBINED) {
            if (has(5)) {
                float t[4];
                tex3(5, t);
                cc = mat.clear_coat * t[0];
                ccpr = mat.clear_coat_roughness * t[1];
            }
        } else {
            float t[4];
            if (has(5)) { tex3(5, t); cc = mat.clear_coat * t[0]; }
            if (has(6)) {
                tex3(6, t);
                ccpr = mat.clear_coat_roughness * ((mflags & R3N_FLAGS_CC_GLTF_SPLIT) ? t[1] : t[0]);
            }
        }
        // --- emissive (opaque.wgsl:395-399); the anisotropy texture (:403-407) feeds nothing downstream
#pragma unroll
        for (int c = 0; c < 3; ++c) px.emissive[c] = mat.emissive[c];
        if (has(7)) {
            float t[4];
            tex3(7, t);
#pragma unroll
            for (int c = 0; c < 3; ++c) px.emissive[c] = mat.emissive[c] * t[c];
        }
        const float omm = 1.0f - metallic;
#pragma unroll
        for (int c = 0; c < 3; ++c) px.diffuse[c] = px.albedo[c] * omm;
        const float refl = (0.16f * reflectance) * reflectance;
        const float refl_omm = refl * omm;
#pragma unroll
        for (int c = 0; c < 3; ++c) px.f0[c] = M::mad(px.albedo[c], metallic, refl_omm);
        if (kClearCoat && cc != 0.0f) {
            const float base_pr = fmaxf(pr, ccpr);
            pr = M::mad(base_pr, cc, pr * (1.0f - cc));
        }
        px.roughness = pr * pr;
        px.ao = ao;

        float vv[3] = {vpos[0], vpos[1], vpos[2]};
        normalize3m<M>(vv);
#pragma unroll
        for (int c = 0; c < 3; ++c) vv[c] = -vv[c];
        float color[3] = {px.emissive[0], px.emissive[1], px.emissive[2]};
        // A fully occluded light (shadow * ao == 0) adds (finite) * 0 = +-0 when every factor of surface_shading is
        // finite: skip its BRDF.  Finite is guaranteed by: all pixel inputs finite (the sum of magnitudes is finite;
        // NaN fails the comparison), roughness^2 >= 1e-9 (D <= 1/(pi a^2) <= 3.2e17 without underflow of f^2,
        // V <= 0.5/(1e-5 a) <= 5e13, Fresnel <= 2) and |light colour| <= 1e6 (checked when the lights are staged).
        const float mag = (((fabsf(px.normal[0]) + fabsf(px.normal[1])) + (fabsf(px.normal[2]) + fabsf(vv[0]))) +
                           ((fabsf(vv[1]) + fabsf(vv[2])) + (fabsf(px.f0[0]) + fabsf(px.f0[1])))) +
                          (((fabsf(px.f0[2]) + fabsf(px.diffuse[0])) + (fabsf(px.diffuse[1]) + fabsf(px.diffuse[2]))) + fabsf(px.ao));
        const bool skip_ok = px.roughness >= 1e-9f && px.roughness <= 1e9f && mag < 1e30f;
        const BrdfPixel pre_v = brdf_pixel<M>(px, vv);
        const BrdfPixel *pre = &pre_v;
        for (uint32_t i = 0; i < n_dir; ++i) {
            const LdsDirLight L = load_light<LdsDirLight, SL>(s_dir + i);
            // surface_shading scales by k = nol * occlusion.  With nol == 0 and roughness > 0 every factor is finite
            // (D <= 1/(pi a^2), V <= 0.5/(nov a), nov >= 1e-5), so the light adds exactly +0: skip the shadow lookup
            // and the BRDF.  `+= 0.0f` keeps the -0 -> +0 behaviour of the full expression.
            const float nl_raw = dot3m<M>(px.normal, L.l);
            if (px.roughness > 0.0f && nl_raw == nl_raw && sat(nl_raw) == 0.0f) {  // (a NaN normal must stay NaN)
#pragma unroll
                for (int c = 0; c < 3; ++c) color[c] += 0.0f;
                continue;
            }
            // shadow coordinates: exact arithmetic under both policies (they select shadow texels and feed a comparison)
            float sn[4];
            mul_vec4m<MathExact>(L.m, vpos[0], vpos[1], vpos[2], vpos[3], sn);
            const float fl[2] = {sn[0] * 0.5f + 0.5f, sn[1] * 0.5f + 0.5f};
            const float local[2] = {fl[0], 1.0f - fl[1]};
            float tl[2] = {L.offset[0], L.offset[1]};
            float tr[2] = {tl[0] + L.size[0], tl[1] + L.size[1]};
            const float coords[2] = {tl[0] * (1.0f - local[0]) + tr[0] * local[0],
                                     tl[1] * (1.0f - local[1]) + tr[1] * local[1]};
            const float border[2] = {L.inv_res[0] * 1.5f, L.inv_res[1] * 1.5f};
            tl[0] += border[0]; tl[1] += border[1];
            tr[0] -= border[0]; tr[1] -= border[1];
            float shadow = 1.0f;
            // opaque.wgsl:509-514 (quirk: `any`, un-atlased coords vs atlas-space bounds -- reproduced)
            if ((fl[0] >= tl[0] || fl[1] >= tl[1]) && (fl[0] <= tr[0] || fl[1] <= tr[1]) && sn[2] >= 0.0f && sn[2] <= 1.0f)
                shadow = shadow_pcf5<M>(a.atlas, a.atlas_w, a.atlas_h, coords[0], coords[1], sn[2]);
            if (skip_ok && L.sane != 0.0f && shadow * px.ao == 0.0f) {
#pragma unroll
                for (int c = 0; c < 3; ++c) color[c] += 0.0f;
                continue;
            }
            float res[3];
            surface_shading<M>(L.l, L.color, px, vv, shadow * px.ao, res, pre);
#pragma unroll
            for (int c = 0; c < 3; ++c) color[c] += res[c];
        }
        for (uint32_t i = 0; i < n_point; ++i) {
            const LdsPointLight P = load_light<LdsPointLight, SL>(s_point + i);
            const float delta[3] = {P.vpos[0] - vpos[0], P.vpos[1] - vpos[1], P.vpos[2] - vpos[2]};
            const float d = M::sqrt(dot3m<M>(delta, delta));
            const float s = sat(M::div(d, P.radius));
            const float s2 = s * s, is2 = 1.0f - s2;
            const float att = M::div(is2 * is2, 1.0f + s2);
            const float inten[3] = {P.color[0] * att, P.color[1] * att, P.color[2] * att};
            const float l[3] = {M::div(delta[0], d), M::div(delta[1], d), M::div(delta[2], d)};
            float res[3];
            surface_shading<M>(l, inten, px, vv, px.ao, res, pre);
#pragma unroll
            for (int c = 0; c < 3; ++c) color[c] += (res[c] > 0.0f ? res[c] : 0.0f);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c] = fmaxf(a.fu->ambient[c] * px.albedo[c], color[c]);
        out[3] = fmaxf(a.fu->ambient[3] * px.albedo[3], px.albedo[3]);
    }
}

template <bool TEX, class M = MathExact>
R3N_DEV void shade_fragment(const ShadeArgs &a, const LdsDirLight *s_dir, const LdsPointLight *s_point, uint32_t n_dir,
                            uint32_t n_point, uint32_t id, uint32_t x, uint32_t y, float out[4]) {
    TriRecord r;
    vertex_stage<TEX>(a, id, r);
    fragment_stage<TEX, M>(a, s_dir, s_point, n_dir, n_point, r, x, y, out);
}

// Flags the triangles that own at least one pixel (plain byte stores: every writer writes 1).
static __global__ __launch_bounds__(256) void k_mark_visible(const unsigned long long *__restrict__ vis, unsigned char *__restrict__ seen,
                                                      size_t first_pixel, size_t n_pixels) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    const uint32_t id = i < n_pixels ? (uint32_t)(vis[first_pixel + i] & 0xFFFFFFFFull) : 0u;
    // consecutive pixels of a row mostly belong to one triangle: only the first lane of a run stores (the lane to the left inside
    // the 16-lane row, DPP row_shr:1; a row's first lane always does) -- 8.3 M byte stores to 130 k distinct bytes became ~1.5 M
    const uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp((int)~id, (int)id, 0x111, 0xF, 0xF, false);
    if (id != 0u && id != left) seen[id - 1u] = 1;
}
// One thread per canonical triangle slot: the vertex stage + setup of the flagged ones, once per frame.
template <bool TEX>
__global__ __launch_bounds__(256) void k_vertex_stage(ShadeArgs a) {
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= a.total_tris || !a.seen[slot]) return;
    a.seen[slot] = 0;  // consumed: the flags are all zero again when the launch ends (no clear between frames)
    TriRecord r;
    vertex_stage<TEX>(a, slot + 1u, r);
    r._pad[0] = r._pad[1] = r._pad[2] = r._pad[3] = 0u;
    a.tri_rec[slot] = r;
}

// The light list in view space, once per workgroup (LDS): matrices light.view_proj * uniforms.inv_view, directions,
// point-light positions.  Ends with a barrier.
R3N_DEV void stage_lights(const ShadeArgs &a, LdsDirLight *s_dir, LdsPointLight *s_point, uint32_t &n_dir, uint32_t &n_point) {
    n_dir = min(*reinterpret_cast<const uint32_t *>(a.dir_buf), (uint32_t)R3N_MAX_DIR_LIGHTS);
    n_point = min(*reinterpret_cast<const uint32_t *>(a.point_buf), (uint32_t)R3N_MAX_POINT_LIGHTS);

    const r3n_dir_light128 *dirs = reinterpret_cast<const r3n_dir_light128 *>(a.dir_buf + 16);
    const r3n_point_light32 *points = reinterpret_cast<const r3n_point_light32 *>(a.point_buf + 16);
    for (uint32_t i = threadIdx.x; i < n_dir * 4u; i += 256u) {
        const uint32_t li = i >> 2, c = i & 3u;
        const float *col = a.fu->inv_view + 4 * c;  // column c of (view_proj * inv_view)
        float o[4];
        mul_vec4(dirs[li].view_proj, col[0], col[1], col[2], col[3], o);
#pragma unroll
        for (int r = 0; r < 4; ++r) s_dir[li].m[4 * c + r] = o[r];
        if (c == 0) {
            const float nd[3] = {-dirs[li].direction[0], -dirs[li].direction[1], -dirs[li].direction[2]};
            float l[3];
            mat3_mul_vec3(a.fu->view, a.fu->view + 4, a.fu->view + 8, nd, l);
            normalize3(l);
#pragma unroll
            for (int r = 0; r < 3; ++r) { s_dir[li].l[r] = l[r]; s_dir[li].color[r] = dirs[li].color[r]; }
            s_dir[li].sane = (fabsf(dirs[li].color[0]) <= 1e6f && fabsf(dirs[li].color[1]) <= 1e6f && fabsf(dirs[li].color[2]) <= 1e6f &&
                              fabsf(l[0]) <= 2.0f && fabsf(l[1]) <= 2.0f && fabsf(l[2]) <= 2.0f) ? 1.0f : 0.0f;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                s_dir[li].inv_res[r] = dirs[li].inv_resolution[r];
                s_dir[li].offset[r] = dirs[li].atlas_offset[r];
                s_dir[li].size[r] = dirs[li].atlas_size[r];
            }
        }
    }
    for (uint32_t i = threadIdx.x; i < n_point; i += 256u) {
        float o[4];
        mul_vec4(a.fu->view, points[i].position[0], points[i].position[1], points[i].position[2], points[i].position[3], o);
#pragma unroll
        for (int r = 0; r < 3; ++r) { s_point[i].vpos[r] = o[r]; s_point[i].color[r] = points[i].color[r]; }
        s_point[i].radius = points[i].radius;
    }
    __syncthreads();
}

// The same once per frame into global memory (one workgroup), for the kernels that read lights through scalar loads.
static __global__ __launch_bounds__(256) void k_stage_view_lights(ShadeArgs a, ViewLights *out) {
    uint32_t n_dir, n_point;
    stage_lights(a, out->dir, out->point, n_dir, n_point);
    if (threadIdx.x == 0u) { out->n_dir = n_dir; out->n_point = n_point; out->_pad[0] = out->_pad[1] = 0u; }
}

// One thread per pixel, 16x16 pixel tiles; the light list is transformed once per workgroup and staged in LDS.
// S = samples per pixel.  S == 4: every sample of the multisampled Rgba16Float target holds the half-rounded colour
// of its nearest fragment (shaded once per distinct triangle, at the pixel centre) or the clear colour; the render
// pass resolve (base.rs:245-258) is their box average ((s0 + s1) + (s2 + s3)) * 0.25.
// Register budget: the untextured single-sample variant is VALU-bound and measurably faster at 5 waves per SIMD
// (<= 96 VGPRs: 347 vs 375 us on the bench scene) -- the second launch-bound asks for that.
// REC: the per-triangle records exist: no vertex-stage code in the kernel at all.
// FAST: the MathFast policy in the fragment stage (opt-in, r3n_config.shade_mode); instantiated for the record-based
// single-sample resolve.
// CLS / VARIANT (S == 1 with records): the material class this instantiation is compiled for and its place in the chain
// (R3N_CLS_*).  With more than one variant in flight (a.variants) a workgroup first ORs its pixels' features and leaves
// unless the tile is its own: the smallest launched variant that covers the tile.
template <int S, bool TEX, bool REC = false, bool SPLIT = false, bool FAST = false, uint32_t CLS = R3N_CLS_ALL, uint32_t VARIANT = 3u>
__global__ __launch_bounds__(256, (S == 1 && !TEX) ? 5 : (REC ? (S == 1 ? (CLS == R3N_CLS_PBR3 ? R3N_BATCH_OCC : R3N_TEX_OCC) : R3N_MS_OCC) : 1)) void k_resolve_opaque(ShadeArgs a) {
    typedef typename std::conditional<FAST, MathFast, MathExact>::type M;
    static_assert(CLS == R3N_CLS_ALL || (S == 1 && REC), "material classes exist for the single-sample record-based resolve");
    __shared__ LdsDirLight s_dir[R3N_MAX_DIR_LIGHTS];
    __shared__ LdsPointLight s_point[R3N_MAX_POINT_LIGHTS];
    __shared__ float s_decode[512];
    // each wavefront shades an 8x8 pixel quad of the 16x16 tile (fewer distinct triangles / atlas texels per wave
    // than a 16x4 strip; measured 3 % faster)
    const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63u;
    const uint32_t bx = blockIdx.x, by = blockIdx.y;
    const uint32_t x = bx * 16u + (ln & 7u) + 8u * (wv & 1u);
    const uint32_t y = a.row_begin + by * 16u + (ln >> 3) + 8u * (wv >> 1);
    const bool inside = x < a.width && y < a.row_end;
    const size_t pix = inside ? (size_t)y * a.width + x : 0u;
    uint32_t id1 = 0u;  // S == 1: the pixel's triangle (canonical slot + 1), 0 = background
    if (S == 1) {
        id1 = inside ? (uint32_t)(a.vis[pix] & 0xFFFFFFFFull) : 0u;
        if (REC && (a.variants & (a.variants - 1u)) != 0u) {  // (launch-uniform) several variants run: is this tile mine?
            __shared__ uint32_t s_feat;
            if (threadIdx.x == 0u) s_feat = 0u;
            __syncthreads();
            uint32_t f = id1 ? a.tri_rec[id1 - 1u].feat : 0u;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) f |= __shfl_xor(f, d);
            if (ln == 0u && f) atomicOr(&s_feat, f);
            __syncthreads();
            // the tile's variant: the smallest launched one at or above the variant of the union (the chain's top is always
            // launched when any material needs it: the census in r3n.hip)
            uint32_t v = r3n_variant_of(s_feat);
            while (v < R3N_VARIANTS - 1u && !((a.variants >> v) & 1u)) ++v;
            if (v != VARIANT) return;  // (workgroup-uniform)
        }
    }
    if (TEX) {  // texel decode tables into LDS (texture.h)
        s_decode[threadIdx.x] = a.tex.decode[threadIdx.x];
        s_decode[256u + threadIdx.x] = a.tex.decode[256u + threadIdx.x];
        a.tex.decode = s_decode;
    }
    // the single-sample record-based resolve reads the frame's light lists from global memory through scalar loads
    // (ViewLights, written by k_stage_view_lights in front of this launch); the other forms stage them per workgroup in LDS
    constexpr bool SL = S == 1 && REC;
    uint32_t n_dir, n_point;
    const LdsDirLight *dirs = s_dir;
    const LdsPointLight *points = s_point;
    if (SL) {
        n_dir = load_light<uint32_t, true>(&a.view_lights->n_dir);
        n_point = load_light<uint32_t, true>(&a.view_lights->n_point);
        dirs = a.view_lights->dir;
        points = a.view_lights->point;
        if (TEX) __syncthreads();  // the decode tables
    } else {
        stage_lights(a, s_dir, s_point, n_dir, n_point);
    }
    if (!inside && !SPLIT) return;  // SPLIT: every thread of the workgroup takes part in the queue reservation below
    float out[4];
    if (S == 1) {
        const uint32_t id = id1;
        if (id == 0u) {
            const ushort4 hc = pack_half4(a.clear);
            a.hdr_out[pix] = hc;
            a.ldr_out[pix] = tonemap_half4(a.srgb_lut, hc, a.out_bgr);
            return;
        }
        if (REC) fragment_stage<TEX, M, CLS, SL>(a, dirs, points, n_dir, n_point, a.tri_rec[id - 1u], x, y, out);
        else shade_fragment<TEX, M>(a, s_dir, s_point, n_dir, n_point, id, x, y, out);
    } else {
        uint32_t ids[S];
        float col[S][4];
#pragma unroll
        for (int sm = 0; sm < S; ++sm) ids[sm] = inside ? (uint32_t)(a.vis[pix * (size_t)S + (size_t)sm] & 0xFFFFFFFFull) : 0u;
        // The distinct triangles among the pixel's samples, each shaded ONCE (same triangle, same pixel centre: same
        // value) by one copy of the fragment stage in a rolled loop: unrolling it per sample made the kernel four
        // fragment stages long (instruction cache, registers) although interior pixels hold one triangle.
        uint32_t first_of[S];  // index of the first sample with the same id
        uint32_t n_unique = 0;
#pragma unroll
        for (int sm = 0; sm < S; ++sm) {
            uint32_t f = (uint32_t)sm;
#pragma unroll
            for (int p = sm - 1; p >= 0; --p)
                if (ids[p] == ids[sm]) f = (uint32_t)p;
            first_of[sm] = f;
            n_unique += f == (uint32_t)sm ? 1u : 0u;
        }
#pragma unroll
        for (int sm = 0; sm < S; ++sm)
#pragma unroll
            for (int c = 0; c < 4; ++c) col[sm][c] = 0.0f;
        // SPLIT: this kernel shades only the triangle of sample 0 -- every lane busy once -- and queues the pixel's
        // other triangles (edge pixels, a minority) for k_resolve_edges, which runs them densely; k_resolve_edge_pixels
        // then averages.  Unsplit, a wavefront pays a whole fragment stage for every extra triangle of its worst pixel.
        uint32_t edge_base = 0;
        bool edge_fits = true;
        if (SPLIT) {
            __shared__ uint32_t s_extra, s_base;
            if (threadIdx.x == 0u) s_extra = 0u;
            __syncthreads();
            const uint32_t extra = n_unique - 1u;
            uint32_t my_off = 0;
            if (extra) my_off = atomicAdd(&s_extra, extra);
            __syncthreads();
            const uint32_t q = (blockIdx.y * gridDim.x + blockIdx.x) % R3N_EDGEQ;
            if (threadIdx.x == 0u && s_extra) s_base = atomicAdd(&a.edge_count[q], s_extra);
            __syncthreads();
            edge_base = s_base + my_off;
            edge_fits = extra == 0u || edge_base + extra <= a.edge_capacity;  // else: shade everything here (never drop work)
            if (extra && !edge_fits) {  // the slots this pixel reserved inside the list stay empty
                for (uint32_t k = edge_base; k < min(edge_base + extra, a.edge_capacity); ++k)
                    a.edge_list[(size_t)q * a.edge_capacity + k] = 0xFFFFFFFFu;
            }
            if (extra && edge_fits) {
                uint32_t *dst = a.edge_list + (size_t)q * a.edge_capacity + edge_base;
                uint32_t last = 0;
#pragma unroll
                for (int sm = 1; sm < S; ++sm)
                    if (first_of[sm] == (uint32_t)sm) last = (uint32_t)sm;
                uint32_t w = 0;
#pragma unroll
                for (int sm = 1; sm < S; ++sm)
                    if (first_of[sm] == (uint32_t)sm) dst[w++] = ((uint32_t)pix << 3) | ((uint32_t)sm << 1) | (last == (uint32_t)sm ? 1u : 0u);
            }
        }
        if (!inside) return;  // (after the workgroup barriers)
        if (SPLIT && edge_fits) {
            // the common case, kept lean: only the first triangle is shaded here, so nothing per sample has to stay in
            // registers across the fragment stage except which samples it owns
            uint32_t mask0 = 0;
#pragma unroll
            for (int sm = 0; sm < S; ++sm) mask0 |= first_of[sm] == 0u ? 1u << sm : 0u;
            const uint32_t id0 = ids[0];
            const bool single = n_unique == 1u;
            float v[4];
            if (id0 == 0u) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = a.clear[c];
            } else if (REC) {
                fragment_stage<TEX, M>(a, s_dir, s_point, n_dir, n_point, a.tri_rec[id0 - 1u], x, y, v);
            } else {
                shade_fragment<TEX, M>(a, s_dir, s_point, n_dir, n_point, id0, x, y, v);
            }
            const ushort4 h = pack_half4(v);
            if (!single) {  // edge pixel: park the samples of the first triangle; the other passes finish the pixel
#pragma unroll
                for (int sm = 0; sm < S; ++sm)
                    if ((mask0 >> sm) & 1u) a.samples_out[pix * (size_t)S + (size_t)sm] = h;
                return;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {  // box resolve of four equal samples, same expression as everywhere
                const float cf = (float)(_Float16)v[c];
                out[c] = ((cf + cf) + (cf + cf)) * 0.25f;
            }
            const ushort4 ho = pack_half4(out);
            a.hdr_out[pix] = ho;
            a.ldr_out[pix] = tonemap_half4(a.srgb_lut, ho, a.out_bgr);
            return;
        }
        const uint32_t n_here = n_unique;
#pragma unroll 1
        for (uint32_t k = 0, sm_at = 0; k < n_here; ++k, ++sm_at) {
            // next leader sample.  BOUNDED on purpose: there is always a leader at or behind sm_at (n_here counts them), so the bound never
            // binds in a correct execution -- but a build of this kernel (round 6: a six-line change elsewhere in the fragment stage,
            // -O3, occupancy target 5) spun in exactly this loop on the device until the process was killed, returned with the bound in
            // place, and produced the oracle's bits (profiles/r06_native_hang.md): a search that cannot leave [0, S) cannot hang the GPU
            while (sm_at < (uint32_t)(S - 1) && first_of[sm_at == 0u ? 0 : (sm_at == 1u ? 1 : (sm_at == 2u ? 2 : 3))] != sm_at) ++sm_at;
            const uint32_t id = sm_at == 0u ? ids[0] : (sm_at == 1u ? ids[1] : (sm_at == 2u ? ids[2] : ids[3]));
            float v[4];
            if (id == 0u) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = a.clear[c];
            } else if (REC) {
                fragment_stage<TEX, M>(a, s_dir, s_point, n_dir, n_point, a.tri_rec[id - 1u], x, y, v);
            } else {
                shade_fragment<TEX, M>(a, s_dir, s_point, n_dir, n_point, id, x, y, v);
            }
#pragma unroll
            for (int sm = 0; sm < S; ++sm)
                if (first_of[sm] == sm_at) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) col[sm][c] = (float)(_Float16)v[c];
                }
        }
        if (!SPLIT && a.samples_out != nullptr) {
#pragma unroll
            for (int sm = 0; sm < S; ++sm) a.samples_out[pix * (size_t)S + (size_t)sm] = pack_half4(col[sm]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = ((col[0][c] + col[1][c]) + (col[2][c] + col[3][c])) * 0.25f;
    }
    const ushort4 ho = pack_half4(out);
    a.hdr_out[pix] = ho;
    a.ldr_out[pix] = tonemap_half4(a.srgb_lut, ho, a.out_bgr);
}

// Split MSAA resolve, pass B: one thread per queued (pixel, leader sample): shade that triangle at the pixel centre and
// park the half-rounded colour in every sample it owns.  Pass C (k_resolve_edge_pixels): the entry flagged as its
// pixel's last one averages the four parked samples -- the same box resolve expression as everywhere else.
template <bool TEX, bool REC>
__global__ __launch_bounds__(256, REC ? R3N_MS_OCC : 1) void k_resolve_edges(ShadeArgs a) {
    __shared__ LdsDirLight s_dir[R3N_MAX_DIR_LIGHTS];
    __shared__ LdsPointLight s_point[R3N_MAX_POINT_LIGHTS];
    __shared__ float s_decode[512];
    if (TEX) {
        s_decode[threadIdx.x] = a.tex.decode[threadIdx.x];
        s_decode[256u + threadIdx.x] = a.tex.decode[256u + threadIdx.x];
        a.tex.decode = s_decode;
    }
    uint32_t n_dir, n_point;
    stage_lights(a, s_dir, s_point, n_dir, n_point);
    const uint32_t q = blockIdx.x % R3N_EDGEQ;
    const uint32_t n = min(a.edge_count[q], a.edge_capacity);
    const uint32_t *list = a.edge_list + (size_t)q * a.edge_capacity;
    const uint32_t stride = (gridDim.x / R3N_EDGEQ) * 256u;
    for (uint32_t i = (blockIdx.x / R3N_EDGEQ) * 256u + threadIdx.x; i < n; i += stride) {
        const uint32_t e = list[i];
        if (e == 0xFFFFFFFFu) continue;  // reserved by a pixel that did not fit and shaded itself
        const size_t pix = e >> 3;
        const uint32_t leader = (e >> 1) & 3u;
        uint32_t ids[4];
#pragma unroll
        for (int sm = 0; sm < 4; ++sm) ids[sm] = (uint32_t)(a.vis[pix * 4u + (size_t)sm] & 0xFFFFFFFFull);
        const uint32_t id = leader == 1u ? ids[1] : (leader == 2u ? ids[2] : ids[3]);
        const uint32_t x = (uint32_t)(pix % a.width), y = (uint32_t)(pix / a.width);
        float v[4];
        if (id == 0u) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = a.clear[c];
        } else if (REC) {
            fragment_stage<TEX>(a, s_dir, s_point, n_dir, n_point, a.tri_rec[id - 1u], x, y, v);
        } else {
            shade_fragment<TEX>(a, s_dir, s_point, n_dir, n_point, id, x, y, v);
        }
        const ushort4 h = pack_half4(v);
#pragma unroll
        for (int sm = 1; sm < 4; ++sm)
            if ((uint32_t)sm >= leader && ids[sm] == id) a.samples_out[pix * 4u + (size_t)sm] = h;  // samples led by `leader`
    }
}
static __global__ __launch_bounds__(256) void k_resolve_edge_pixels(ShadeArgs a) {
    const uint32_t q = blockIdx.x % R3N_EDGEQ;
    const uint32_t n = min(a.edge_count[q], a.edge_capacity);
    const uint32_t *list = a.edge_list + (size_t)q * a.edge_capacity;
    const uint32_t stride = (gridDim.x / R3N_EDGEQ) * 256u;
    for (uint32_t i = (blockIdx.x / R3N_EDGEQ) * 256u + threadIdx.x; i < n; i += stride) {
        const uint32_t e = list[i];
        if (e == 0xFFFFFFFFu || !(e & 1u)) continue;
        const size_t pix = e >> 3;
        float col[4][4], out[4];
#pragma unroll
        for (int sm = 0; sm < 4; ++sm) {
            const ushort4 h = a.samples_out[pix * 4u + (size_t)sm];
            col[sm][0] = (float)__builtin_bit_cast(_Float16, h.x); col[sm][1] = (float)__builtin_bit_cast(_Float16, h.y);
            col[sm][2] = (float)__builtin_bit_cast(_Float16, h.z); col[sm][3] = (float)__builtin_bit_cast(_Float16, h.w);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = ((col[0][c] + col[1][c]) + (col[2][c] + col[3][c])) * 0.25f;
        const ushort4 ho = pack_half4(out);
        a.hdr_out[pix] = ho;
        a.ldr_out[pix] = tonemap_half4(a.srgb_lut, ho, a.out_bgr);
    }
}

// ------------------------------------------------------------------------------------------------ transparent pass
// Stage 3 (row N3): the collected fragments, one linked list per pixel sample.  The thread of a sample with a list applies its
// fragments in DRAW ORDER -- the next one is the node with the smallest order above the last one applied (orders are unique
// inside a sample: a triangle covers it once; lists are a few nodes long) -- evaluate the fragment (once per triangle and pixel
// centre, like the forward pass), BlendState::ALPHA_BLENDING on the half-rounded destination -- rgb = src * a + dst * (1 - a),
// alpha = src.a + dst.a * (1 - a) in f32, result rounded to half -- exactly the oracle's sequence.  Nothing here needs the
// fragment count on the host: the launch covers the samples, workgroups without a fragment leave before staging anything.
template <int S, bool TEX>
__global__ __launch_bounds__(256) void k_blend_apply(ShadeArgs a, BlendApplyArgs b) {
    __shared__ LdsDirLight s_dir[R3N_MAX_DIR_LIGHTS];
    __shared__ LdsPointLight s_point[R3N_MAX_POINT_LIGHTS];
    __shared__ float s_decode[512];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t ps = b.first_sample + i;
    const uint32_t head = i < b.n_samples ? b.head[ps] : R3N_INVALID;
    if (!__syncthreads_or(head != R3N_INVALID)) return;
    if (TEX) {
        s_decode[threadIdx.x] = a.tex.decode[threadIdx.x];
        s_decode[256u + threadIdx.x] = a.tex.decode[256u + threadIdx.x];
        a.tex.decode = s_decode;
    }
    uint32_t n_dir, n_point;
    stage_lights(a, s_dir, s_point, n_dir, n_point);
    if (head == R3N_INVALID) return;
    const uint32_t pix = ps / (uint32_t)S;
    const uint32_t x = pix % a.width, y = pix / a.width;
    const ushort4 d16 = b.samples[ps];
    float d[4] = {(float)__builtin_bit_cast(_Float16, d16.x), (float)__builtin_bit_cast(_Float16, d16.y),
                  (float)__builtin_bit_cast(_Float16, d16.z), (float)__builtin_bit_cast(_Float16, d16.w)};
    bool first = true;
    uint32_t last = 0u;
    // Both loops carry a bound that never binds on a well-formed list (a list has at most `capacity` nodes, every round consumes
    // one): the shape -- a data-dependent loop inside a data-dependent loop around the whole fragment stage -- is the one a build of
    // the multisampled resolve spun in this round (profiles/r06_native_hang.md); a traversal that is bounded cannot hang the GPU.
    for (uint32_t round = 0; round < b.capacity; ++round) {
        uint32_t best = R3N_INVALID, best_order = 0xFFFFFFFFu;
        uint32_t steps = 0;
        for (uint32_t n = head; n != R3N_INVALID && steps < b.capacity; ++steps) {
            const unsigned long long k = b.keys[n];
            const uint32_t order = (uint32_t)k;
            if ((first || order > last) && order <= best_order) { best = n; best_order = order; }
            n = (uint32_t)(k >> 32);
        }
        if (best == R3N_INVALID) break;
        first = false;
        last = best_order;
        float src[4];
        shade_fragment<TEX>(a, s_dir, s_point, n_dir, n_point, b.vals[best], x, y, src);
        const float al = src[3];
        float r[4];
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c] = src[c] * al + d[c] * (1.0f - al);
        r[3] = src[3] * 1.0f + d[3] * (1.0f - al);
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = (float)(_Float16)r[c];
        if (last == 0xFFFFFFFFu) break;
    }
    b.samples[ps] = pack_half4(d);
}

// Render-pass resolve of the blended samples (S == 4): box average, same expression as in k_resolve_opaque.
static __global__ __launch_bounds__(256) void k_resolve_samples(const ushort4 *__restrict__ samples, ushort4 *__restrict__ hdr_out,
                                                         size_t first_pixel, size_t n_pixels) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n_pixels) return;
    const size_t pix = first_pixel + i;
    float out[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float col[4][4];
#pragma unroll
    for (int sm = 0; sm < 4; ++sm) {
        const ushort4 h = samples[pix * 4u + (size_t)sm];
        col[sm][0] = (float)__builtin_bit_cast(_Float16, h.x); col[sm][1] = (float)__builtin_bit_cast(_Float16, h.y);
        col[sm][2] = (float)__builtin_bit_cast(_Float16, h.z); col[sm][3] = (float)__builtin_bit_cast(_Float16, h.w);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) out[c] = ((col[0][c] + col[1][c]) + (col[2][c] + col[3][c])) * 0.25f;
    hdr_out[pix] = pack_half4(out);
}

// ------------------------------------------------------------------------------------------------ K7 tonemap
// 2 pixels per thread: one 16-byte load, one 8-byte store.
static __global__ __launch_bounds__(256) void k_tonemap(const ushort4 *__restrict__ hdr, uchar4 *__restrict__ out,
                                                 float4 *__restrict__ out_f32, size_t first_pixel, size_t n_pixels,
                                                 const unsigned char *__restrict__ srgb_lut, uint32_t output_format) {
    const bool bgr = (output_format & 1u) != 0u, manual = (output_format & 2u) != 0u;
    const size_t pair = (size_t)blockIdx.x * 256u + threadIdx.x;
    const size_t i0 = first_pixel + pair * 2u;
    if (pair * 2u >= n_pixels) return;
    const bool two = pair * 2u + 1u < n_pixels;
    ushort4 h[2];
    if (two && (i0 & 1u) == 0u) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(hdr + i0);
        h[0] = make_ushort4(raw.x & 0xFFFFu, raw.x >> 16, raw.y & 0xFFFFu, raw.y >> 16);
        h[1] = make_ushort4(raw.z & 0xFFFFu, raw.z >> 16, raw.w & 0xFFFFu, raw.w >> 16);
    } else {
        h[0] = hdr[i0];
        h[1] = two ? hdr[i0 + 1u] : make_ushort4(0, 0, 0, 0);
    }
    uchar4 o8[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        o8[k] = tonemap_half4(srgb_lut, h[k], bgr);
        if (out_f32 != nullptr && (k == 0 || two)) {  // float view of the same target (readback tap only)
            const float r = (float)__builtin_bit_cast(_Float16, h[k].x), g = (float)__builtin_bit_cast(_Float16, h[k].y);
            const float b = (float)__builtin_bit_cast(_Float16, h[k].z), al = (float)__builtin_bit_cast(_Float16, h[k].w);
            out_f32[i0 + (size_t)k] = make_float4(manual ? srgb_scene_to_display(r) : srgb_oetf(r), manual ? srgb_scene_to_display(g) : srgb_oetf(g),
                                                  manual ? srgb_scene_to_display(b) : srgb_oetf(b),
                                                  (!(al > 0.0f)) ? 0.0f : (al >= 1.0f ? 1.0f : al));
        }
    }
    out[i0] = o8[0];
    if (two) out[i0 + 1u] = o8[1];
}
#endif  // R3N_SHADE_DECL_ONLY

// ------------------------------------------------------------------------------------------------ launchers (shade.hip)
// Each returns the hipError_t of the launch as an int.  They only enqueue; ordering, timing spans and buffer sizing stay in r3n.hip.
extern "C" {
int r3n_internal_build_srgb_lut(unsigned char *lut, hipStream_t stream);
// k_mark_visible over keys [first_key, first_key + n_keys) + k_vertex_stage over a.total_tris slots
int r3n_internal_shade_prepass(const ShadeArgs *a, int tex, size_t first_key, size_t n_keys, hipStream_t stream);
// the resolve of rows [a.row_begin, a.row_end): samples 1 | 4; rec = a.tri_rec holds this frame's records; split = three-pass MSAA resolve
// fast: R3N_SHADE_FAST (honoured by the single-sample record-based resolve; the other variants always run the exact arithmetic)
// single-sample record-based resolve: one launch per variant in a->variants (0: the general kernel alone)
int r3n_internal_resolve(const ShadeArgs *a, uint32_t samples, int tex, int rec, int split, int fast, hipStream_t stream);
int r3n_internal_resolve_class(const ShadeArgs *a, uint32_t variant, int fast, hipStream_t stream);  // shade_cls.hip: PLAIN / ALBEDO / PBR3, textured worlds
int r3n_internal_blend_apply(const ShadeArgs *a, const BlendApplyArgs *b, uint32_t samples, int tex, hipStream_t stream);
int r3n_internal_resolve_samples(const ushort4 *samples, ushort4 *hdr_out, size_t first_pixel, size_t n_pixels, hipStream_t stream);
int r3n_internal_tonemap(const ushort4 *hdr, uchar4 *out, float4 *out_f32, size_t first_pixel, size_t n_pixels, const unsigned char *srgb_lut,
                         uint32_t output_format, hipStream_t stream);
}
