/*
 * drt_oracle.c -- TEST INFRASTRUCTURE ONLY (see drt_oracle.h).
 *
 * fp64 restatement of the reference hot path.  The reference builds a per-sample autodiff
 * graph by recursion (pathtracer.hpp:121-136, vector.hpp:488-557) and walks it backwards
 * (vector.hpp:418-484); this file unrolls the same arithmetic, IN THE SAME OPERATION ORDER,
 * into a per-path vertex list with a reverse sweep for the radiance and a forward sweep for
 * the adjoints, so results are bit-identical to the reference (same libm, no FP contraction).
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 */
#include "drt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double v[3]; } v3;

/* ---- Vector<T,3> algebra, vector.hpp:18-118, 327-370, 573-606 ---------------------------- */
static v3 v3_make(double x, double y, double z) { v3 r = {{x, y, z}}; return r; }
static v3 v3_add(v3 a, v3 b) { return v3_make(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]); }
static v3 v3_sub(v3 a, v3 b) { return v3_make(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]); }
static v3 v3_mul(v3 a, v3 b) { return v3_make(a.v[0] * b.v[0], a.v[1] * b.v[1], a.v[2] * b.v[2]); }
/* vector.hpp:95-100: x * s for either operand order */
static v3 v3_scale(v3 a, double s) { return v3_make(a.v[0] * s, a.v[1] * s, a.v[2] * s); }
/* vector.hpp:109-114 */
static v3 v3_div(v3 a, double s) { return v3_make(a.v[0] / s, a.v[1] / s, a.v[2] / s); }
/* vector.hpp:320-325: unary minus is -1 * v */
static v3 v3_neg(v3 a) { return v3_scale(a, -1.0); }
/* vector.hpp:573-578: accumulate(tmp, T()) = ((0 + x) + y) + z */
static double v3_dot(v3 a, v3 b)
{
    v3 t = v3_mul(a, b);
    double s = 0.0;
    s = s + t.v[0];
    s = s + t.v[1];
    s = s + t.v[2];
    return s;
}
/* vector.hpp:580-590 */
static double v3_norm(v3 a) { return sqrt(v3_dot(a, a)); }
static v3 v3_normalize(v3 a) { return v3_div(a, v3_norm(a)); }
/* vector.hpp:592-600 */
static v3 v3_cross(v3 a, v3 b)
{
    return v3_make(a.v[1] * b.v[2] - a.v[2] * b.v[1],
                   a.v[2] * b.v[0] - a.v[0] * b.v[2],
                   a.v[0] * b.v[1] - a.v[1] * b.v[0]);
}
/* vector.hpp:602-606: -v + 2*dot(n, v)*n */
static v3 v3_reflect(v3 v, v3 n)
{
    return v3_add(v3_neg(v), v3_scale(n, 2 * v3_dot(n, v)));
}

static const double k_pi = 3.14159265358979323846; /* constants.hpp:9 */

/* ---- RNG, random.hpp:7-10 ---------------------------------------------------------------- */
typedef struct {
    int mode;
    drt_rng_key path_key;
    uint32_t draw;
    uint64_t extreme;   /* draws that came out as exactly 0 or RAND_MAX */
} rng_t;

static double rng_uniform(rng_t* r)
{
    if (r->mode == DRT_ORACLE_RNG_LIBC)
        return (double)rand() / RAND_MAX;
    {
        const uint32_t v = drt_rng_draw(r->path_key, r->draw++);
        if (v == 0u || v == 2147483647u)
            r->extreme++;           /* uniform() exactly 0 or 1: the samplers' and the roulette's singular points */
        return (double)v / 2147483647.0;
    }
}

uint32_t drt_oracle_rng_u31(uint32_t seed, uint64_t path, uint32_t n)
{
    return drt_rng_u31(seed, path, n);
}

/* ---- shapes, shape.hpp:49-59 (Plane), 78-106 (Sphere) ------------------------------------ */
/* ---- caller-defined kinds (DRT_SHAPE_USER): this CHECKER knows the two test kinds by name -- "disc" and "box", the Disc and
 * AABox plugins of oracle/ref_harness.cpp restated operation for operation (the product compiles the caller's own source) ---- */
static const drt_scene_desc* g_scene;     /* (set by drt_oracle_render: the kinds' names and records 4..7) */
static int user_kind_is(const drt_shape_desc* s, const char* name)
{
    return s->type == DRT_SHAPE_USER && g_scene && s->mesh >= 0 && s->mesh < g_scene->n_kinds && g_scene->kinds[s->mesh].name &&
           strcmp(g_scene->kinds[s->mesh].name, name) == 0;
}
static void user_record(const drt_shape_desc* s, double p[8])
{
    const long i = (long)(s - g_scene->shapes);
    for (int j = 0; j < 4; ++j) {
        p[j] = s->p[j];
        p[4 + j] = g_scene->user_params ? g_scene->user_params[i * 4 + j] : 0.0;
    }
}
static int user_intersect(const drt_shape_desc* s, v3 orig, v3 dir, double* t)
{
    double p[8];
    user_record(s, p);
    if (user_kind_is(s, "disc")) {
        v3 c = v3_make(p[0], p[1], p[2]), n = v3_make(p[3], p[4], p[5]);
        double den = v3_dot(dir, n);
        if (den == 0)
            return 0;
        *t = v3_dot(v3_sub(c, orig), n) / den;
        if (!(*t > 0))
            return 0;
        v3 q = v3_sub(v3_add(orig, v3_scale(dir, *t)), c);
        return v3_dot(q, q) <= p[6] * p[6];
    }
    /* box */
    const double lo[3] = {p[0], p[1], p[2]}, hi[3] = {p[3], p[4], p[5]}, oo[3] = {orig.v[0], orig.v[1], orig.v[2]}, dd[3] = {dir.v[0], dir.v[1], dir.v[2]};
    double tn = -1e300, tf = 1e300;
    for (int a = 0; a < 3; ++a) {
        const double t1 = (lo[a] - oo[a]) / dd[a], t2 = (hi[a] - oo[a]) / dd[a];
        const double ta = t1 < t2 ? t1 : t2, tb = t1 < t2 ? t2 : t1;
        tn = ta > tn ? ta : tn;
        tf = tb < tf ? tb : tf;
    }
    if (!(tn <= tf))
        return 0;
    *t = tn > 0 ? tn : tf;
    return *t > 0;
}
static v3 user_normal(const drt_shape_desc* s, v3 point)
{
    double p[8];
    user_record(s, p);
    if (user_kind_is(s, "disc"))
        return v3_make(p[3], p[4], p[5]);
    const double lo[3] = {p[0], p[1], p[2]}, hi[3] = {p[3], p[4], p[5]}, pp[3] = {point.v[0], point.v[1], point.v[2]};
    int axis = 0;
    double sign = -1, best = fabs(pp[0] - lo[0]);
    for (int a = 0; a < 3; ++a) {
        const double dl = fabs(pp[a] - lo[a]), dh = fabs(pp[a] - hi[a]);
        if (dl < best) { best = dl; axis = a; sign = -1; }
        if (dh < best) { best = dh; axis = a; sign = 1; }
    }
    return v3_make(axis == 0 ? sign : 0.0, axis == 1 ? sign : 0.0, axis == 2 ? sign : 0.0);
}

static int shape_intersect(const drt_shape_desc* s, v3 orig, v3 dir, double* t)
{
    if (s->type == DRT_SHAPE_USER)
        return user_intersect(s, orig, dir, t);
    if (s->type == DRT_SHAPE_PLANE) {
        v3 n = v3_make(s->p[0], s->p[1], s->p[2]);
        double h = v3_dot(orig, n) - s->p[3];
        *t = h / v3_dot(dir, v3_neg(n));
        return *t > 0;
    } else {
        v3 c = v3_make(s->p[0], s->p[1], s->p[2]);
        double radius = s->p[3];
        orig = v3_sub(orig, c);
        double a = 1;
        double b = 2 * v3_dot(orig, dir);
        double cc = v3_dot(orig, orig) - radius * radius;
        double d = b * b - 4 * a * cc;
        if (d < 0)
            return 0;
        double t1 = (-b - sqrt(d)) / (2 * a);
        double t2 = (-b + sqrt(d)) / (2 * a);
        if (t1 > 0 && t2 > 0) {
            *t = t2 < t1 ? t2 : t1; /* std::min(t1, t2) */
            return 1;
        } else if (t1 > 0) {
            *t = t1;
            return 1;
        } else if (t2 > 0) {
            *t = t2;
            return 1;
        }
        return 0;
    }
}

static v3 shape_normal(const drt_shape_desc* s, v3 point)
{
    if (s->type == DRT_SHAPE_USER)
        return user_normal(s, point);
    if (s->type == DRT_SHAPE_PLANE)
        return v3_make(s->p[0], s->p[1], s->p[2]);
    return v3_normalize(v3_sub(point, v3_make(s->p[0], s->p[1], s->p[2])));
}

/* ---- triangles: EXTENSION (the reference has none).  Restates the brute-force
 * `Triangle : drt::Shape<double>` of oracle/ref_harness.cpp operation for operation: two-sided
 * Moller-Trumbore on (v0, e1 = v1 - v0, e2 = v2 - v0), hit iff t > 0, normal =
 * normalize(cross(e1, e2)). */
static void tri_load(const drt_mesh_desc* m, int tri, v3* v0, v3* e1, v3* e2)
{
    const uint32_t* ix = &m->indices[(size_t)tri * 3];
    const double* a = &m->vertices[(size_t)ix[0] * 3];
    const double* b = &m->vertices[(size_t)ix[1] * 3];
    const double* c = &m->vertices[(size_t)ix[2] * 3];
    *v0 = v3_make(a[0], a[1], a[2]);
    *e1 = v3_sub(v3_make(b[0], b[1], b[2]), *v0);
    *e2 = v3_sub(v3_make(c[0], c[1], c[2]), *v0);
}

static int tri_intersect(v3 v0, v3 e1, v3 e2, v3 orig, v3 dir, double* t)
{
    v3 pvec = v3_cross(dir, e2);
    double det = v3_dot(e1, pvec);
    if (det == 0)
        return 0;
    double inv = 1 / det;
    v3 tvec = v3_sub(orig, v0);
    double u = v3_dot(tvec, pvec) * inv;
    if (u < 0 || u > 1)
        return 0;
    v3 qvec = v3_cross(tvec, e1);
    double v = v3_dot(dir, qvec) * inv;
    if (v < 0 || u + v > 1)
        return 0;
    *t = v3_dot(e2, qvec) * inv;
    return *t > 0;
}

/* ---- Pathtracer::raycast, pathtracer.hpp:72-89 ------------------------------------------- */
typedef struct {
    int shape;      /* index into scene->shapes, -1 = miss */
    int tri;        /* triangle within the mesh, -1 for analytic shapes */
    int flat;       /* position in the flattened scene (every triangle counts as a shape) */
    int material;   /* resolved material index (per-face or the shape's), -1 = none */
    int param;      /* colour parameter of the hit's BxDF: the face's own (drt_mesh_desc::face_param) or its material's; -1 = none */
} hit_t;

static hit_t raycast(const drt_scene_desc* sc, v3 orig, v3 dir, v3* point, v3* normal, double* t_out)
{
    double tmin = INFINITY;
    hit_t hit = {-1, -1, -1, -1, -1};
    int flat = 0;
    for (int i = 0; i < sc->n_shapes; ++i) {
        const drt_shape_desc* sh = &sc->shapes[i];
        if (sh->type == DRT_SHAPE_MESH) {
            const drt_mesh_desc* m = &sc->meshes[sh->mesh];
            for (int k = 0; k < m->n_triangles; ++k, ++flat) {
                v3 v0, e1, e2;
                double t;
                tri_load(m, k, &v0, &e1, &e2);
                if (!tri_intersect(v0, e1, e2, orig, dir, &t) || t >= tmin)
                    continue;
                tmin = t;
                *point = v3_add(orig, v3_scale(dir, t));
                *normal = v3_normalize(v3_cross(e1, e2));
                hit.shape = i; hit.tri = k; hit.flat = flat;
                hit.material = m->face_material ? m->face_material[k] : sh->material;
                hit.param = hit.material < 0 ? -1 : (m->face_param && m->face_param[k] >= 0 && sc->materials[hit.material].type != DRT_BXDF_MIRROR
                                                         ? m->face_param[k] : sc->materials[hit.material].param);
            }
            continue;
        }
        double t;
        if (!shape_intersect(sh, orig, dir, &t) || t >= tmin) {
            ++flat;
            continue;
        }
        tmin = t;
        *point = v3_add(orig, v3_scale(dir, t));
        *normal = shape_normal(sh, *point);
        hit.shape = i; hit.tri = -1; hit.flat = flat; hit.material = sh->material;
        hit.param = sh->material >= 0 ? sc->materials[sh->material].param : -1;
        ++flat;
    }
    *t_out = tmin;
    if (isinf(tmin))
        hit.shape = -1;
    return hit;
}

/* ---- make_frame / angle_to_dir, bxdf.hpp:29-52 ------------------------------------------- */
static void make_frame(v3 normal, v3 frame[3])
{
    v3 e1 = v3_make(1., 0., 0.);
    v3 e2 = v3_make(0., 1., 0.);
    v3 tangent;
    if (fabs(v3_dot(e1, normal)) < fabs(v3_dot(e2, normal)))
        tangent = v3_normalize(v3_sub(e1, v3_scale(normal, v3_dot(e1, normal))));
    else
        tangent = v3_normalize(v3_sub(e2, v3_scale(normal, v3_dot(e2, normal))));
    v3 bitangent = v3_normalize(v3_cross(normal, tangent));
    frame[0] = tangent;
    frame[1] = bitangent;
    frame[2] = normal;
}

static v3 angle_to_dir(double theta, double phi, const v3 frame[3])
{
    double x = cos(phi) * sin(theta);
    double y = sin(phi) * sin(theta);
    double z = cos(theta);
    return v3_add(v3_add(v3_scale(frame[0], x), v3_scale(frame[1], y)), v3_scale(frame[2], z));
}

/* ---- BxDF::sample, bxdf.hpp:69-79 (Diffuse), 106-120 (Specular); null: pathtracer.hpp:17-27 */
static v3 bxdf_sample(const drt_material_desc* m, v3 normal, v3 dir_in, rng_t* rng, double* pdf)
{
    if (!m) {
        *pdf = 1;
        return v3_make(0, 0, 0);
    }
    v3 frame[3];
    if (m->type >= DRT_BXDF_USER) {
        /* a caller-defined kind: this CHECKER knows the test kind "coslobe" (the CosLobeBxDF plugin of oracle/ref_harness.cpp restated):
         * a power-cosine lobe around the normal, cos(theta) = u1^(1 / (k + 1)), pdf = (k + 1) / (2 pi) cos^k */
        double k = m->exponent;
        double cos_t = pow(rng_uniform(rng), 1 / (k + 1));
        double theta = acos(cos_t);
        double phi = 2 * k_pi * rng_uniform(rng);
        make_frame(normal, frame);
        v3 dir = angle_to_dir(theta, phi, frame);
        *pdf = (k + 1) / (2 * k_pi) * pow(cos_t, k);
        return dir;
    }
    if (m->type == DRT_BXDF_MIRROR) {
        /* bxdf.hpp:137-142 (reflect(dir_in, normal), pdf 1).  Convention of this build (harness
         * plugin, host API, oracle, device): EVERY BxDF sample advances the stream by two draws, so a
         * draw's position is a closed form of the depth; the mirror discards its two. */
        (void)rng_uniform(rng);
        (void)rng_uniform(rng);
        *pdf = 1;
        return v3_reflect(dir_in, normal);
    }
    if (m->type == DRT_BXDF_DIFFUSE) {
        double theta = asin(sqrt(rng_uniform(rng)));
        double phi = 2 * k_pi * rng_uniform(rng);
        make_frame(normal, frame);
        v3 dir = angle_to_dir(theta, phi, frame);
        *pdf = cos(theta) / k_pi;
        return dir;
    } else {
        double e = m->exponent;
        double theta = acos(sqrt(pow(rng_uniform(rng), 2 / (e + 2))));
        double phi = 2 * k_pi * rng_uniform(rng);
        make_frame(normal, frame);
        v3 halfway = angle_to_dir(theta, phi, frame);
        if (v3_dot(halfway, dir_in) < 0)
            halfway = v3_reflect(halfway, normal);
        v3 dir = v3_reflect(dir_in, halfway);
        *pdf = (e + 2) / (2 * k_pi) * pow(cos(theta), e + 1) * sin(theta);
        return dir;
    }
}

/* BxDF::operator(): value = scale_kind(color): Diffuse color / pi (bxdf.hpp:63-67), Specular
 * factor * color (bxdf.hpp:91-104).  Returns the scalar; *is_div says how it is applied. */
static double bxdf_scalar(const drt_material_desc* m, v3 normal, v3 dir_in, v3 dir_out, int* is_div)
{
    if (m->type == DRT_BXDF_DIFFUSE) {
        *is_div = 1;
        return k_pi;
    }
    if (m->type == DRT_BXDF_MIRROR) {
        *is_div = 1;                      /* bxdf.hpp:133-135: 1 / cos_theta, on every channel */
        return v3_dot(normal, dir_out);
    }
    if (m->type >= DRT_BXDF_USER) {       /* "coslobe": colour * (k + 2) / (2 pi) cos^k(theta_out), 0 below the surface */
        double c = v3_dot(normal, dir_out);
        *is_div = 0;
        return c > 0 ? (m->exponent + 2) / (2 * k_pi) * pow(c, m->exponent) : 0.0;
    }
    v3 halfway = v3_normalize(v3_add(dir_in, dir_out));
    double cos_theta = v3_dot(normal, halfway);
    double sin_theta = sqrt(1 - cos_theta * cos_theta);
    double factor = (m->exponent + 2) / (2 * k_pi) * pow(cos_theta, m->exponent) * sin_theta;
    *is_div = 0;
    return factor;
}

/* ---- per-path vertex list ------------------------------------------------------------------ */
typedef struct {
    double p;        /* roulette survival probability of this depth, pathtracer.hpp:130 */
    double q;        /* pdf of the sampled direction */
    double c;        /* cos_theta = dot(normal, dir_out), pathtracer.hpp:103 */
    double bscalar;  /* pi (Diffuse, divide) or factor (Specular, multiply) */
    int bdiv;
    int color_param; /* -1 = no BxDF */
    int emis_param;  /* -1 = no emitter */
    int material;    /* index into scene->materials, -1 = none */
    v3 f;            /* BxDF value */
    v3 lnext;        /* radiance returned by the recursive trace, filled by the reverse sweep */
    v3 point, normal, dir_in;   /* geometry of the vertex (the unbiased backward re-samples here) */
} vertex_t;

#define ORACLE_MAX_VERTICES 4096

/* Camera::sample, camera.hpp:51-60 */
static v3 camera_sample(const drt_camera_desc* cam, int x, int y, rng_t* rng)
{
    double width = (double)cam->width, height = (double)cam->height;
    double s = (x + rng_uniform(rng)) / width;
    double t = (y + rng_uniform(rng)) / height;
    double aspect = width / height;
    v3 fwd = v3_make(cam->forward[0], cam->forward[1], cam->forward[2]);
    v3 right = v3_make(cam->right[0], cam->right[1], cam->right[2]);
    v3 up = v3_make(cam->up[0], cam->up[1], cam->up[2]);
    v3 dir = fwd;
    dir = v3_add(dir, v3_scale(right, (2. * s - 1.) * aspect * tan(cam->vfov / 2.)));
    dir = v3_add(dir, v3_scale(v3_neg(up), (2. * t - 1.) * tan(cam->vfov / 2.)));
    return v3_normalize(dir);
}

static int32_t g_gimg_param = -1;
static double* g_gimg_out = NULL;
void drt_oracle_set_gradient_image(int32_t param, double* out)
{
    g_gimg_param = out ? param : -1;
    g_gimg_out = out;
}

typedef struct {
    const drt_scene_desc* scene;
    const drt_render_params* rp;
    rng_t rng;
    int faithful, zero_dir_miss, max_depth;
    drt_oracle_stats st;
    drt_oracle_vertex* vertices;
    uint64_t max_vertices, nvlog;
    int logging, log_ord;   /* log_ord: ordinal of the raycast within the camera sample */
    double log_path;
    int status;
} walk_ctx;

static v3 param_rgb(const drt_scene_desc* sc, int p)
{
    return v3_make(sc->params[p * 3], sc->params[p * 3 + 1], sc->params[p * 3 + 2]);
}

/* BxDF value at a vertex for a given outgoing direction (fills f, bscalar, bdiv, c) */
static void vertex_eval(const drt_scene_desc* sc, vertex_t* v, v3 dir_out)
{
    const drt_material_desc* m = v->material >= 0 ? &sc->materials[v->material] : NULL;
    if (m) {
        v->bscalar = bxdf_scalar(m, v->normal, v->dir_in, dir_out, &v->bdiv);
        v3 color = v->color_param >= 0 ? param_rgb(sc, v->color_param) : v3_make(1, 1, 1);   /* mirror: no colour */
        v->f = v->bdiv ? v3_div(color, v->bscalar) : v3_scale(color, v->bscalar);
    } else {
        v->bscalar = 0;
        v->bdiv = 0;
        v->f = v3_make(0, 0, 0); /* pathtracer.hpp:38-39 */
    }
    v->c = v3_dot(v->normal, dir_out);
}

/* Pathtracer::trace / scatter (pathtracer.hpp:91-136) from (orig, dir) at `depth`: the forward
 * walk fills vtx[0..nv) and the reverse sweep
 *   L_k = (E + (0 + ((f*L_{k+1})*c)/q)) / p
 * (pathtracer.hpp:104 -> vector.hpp:515,532; integrate.hpp:31,34 -> vector.hpp:553,493;
 * pathtracer.hpp:114,133) leaves the radiance of the path in *L. Returns nv. */
static int walk(walk_ctx* w, v3 orig, v3 dir, int depth, vertex_t* vtx, v3* L)
{
    const drt_scene_desc* scene = w->scene;
    const drt_render_params* rp = w->rp;
    int nv = 0;
    for (;;) {
        /* extension, not in the reference: a user cap is tested BEFORE the roulette (no draw at the cap) -- unless the roulette
         * of this depth ends the path with certainty anyway (absorb == 1 at or beyond min_bounces): then the cap cuts nothing,
         * the draw is consumed as the reference consumes it, and the render is the reference's (drt_hip.h: max_depth) */
        if (w->max_depth && depth >= w->max_depth && !(rp->absorb >= 1.0 && depth >= rp->min_bounces))
            break;
        if (depth >= rp->min_bounces && rng_uniform(&w->rng) < rp->absorb)
            break;
        double p = depth >= rp->min_bounces ? (1 - rp->absorb) : 1;
        v3 point = v3_make(0, 0, 0), normal = v3_make(0, 0, 0);
        double t;
        int zero_dir = dir.v[0] == 0 && dir.v[1] == 0 && dir.v[2] == 0;
        hit_t hit;
        if (zero_dir && w->zero_dir_miss) {
            hit.shape = -1; hit.tri = -1; hit.flat = -1; hit.material = -1; hit.param = -1;
            t = INFINITY;
        } else {
            hit = raycast(scene, orig, dir, &point, &normal, &t);
        }
        const int shape = hit.shape;
        if (zero_dir) w->st.zero_dir_segments++; else w->st.segments++;
        if (!zero_dir && (uint64_t)depth + 1 > w->st.deepest) w->st.deepest = (uint64_t)depth + 1;
        if (w->logging && w->nvlog < w->max_vertices) {
            drt_oracle_vertex* Lg = &w->vertices[w->nvlog++];
            memset(Lg, 0, sizeof *Lg);
            Lg->path = w->log_path;
            Lg->depth = w->log_ord++;   /* = depth in the forward pass; keeps counting in backward */
            for (int c = 0; c < 3; ++c) { Lg->o[c] = orig.v[c]; Lg->d[c] = dir.v[c]; }
            Lg->shape = shape >= 0 ? hit.flat : -1;
            if (shape >= 0) {
                Lg->t = t;
                for (int c = 0; c < 3; ++c) { Lg->p[c] = point.v[c]; Lg->n[c] = normal.v[c]; }
            }
        }
        if (shape < 0)
            break;
        if (nv >= ORACLE_MAX_VERTICES) {
            w->status = DRT_ERR_INVALID;
            break;
        }
        const drt_shape_desc* sh = &scene->shapes[shape];
        const drt_material_desc* m = hit.material >= 0 ? &scene->materials[hit.material] : NULL;
        vertex_t* v = &vtx[nv++];
        v->p = p;
        v->emis_param = sh->emitter >= 0 ? scene->emitters[sh->emitter].param : -1;
        v->color_param = m ? hit.param : -1;
        v->material = hit.material;
        v->point = point;
        v->normal = normal;
        v->dir_in = v3_neg(dir);
        v3 dir_out = bxdf_sample(m, normal, v->dir_in, &w->rng, &v->q);
        vertex_eval(scene, v, dir_out);
        if (!m && !w->faithful)
            break; /* the continuation contributes exactly 0 */
        orig = v3_add(point, v3_scale(dir_out, 1e-3)); /* pathtracer.hpp:99 */
        dir = dir_out;
        ++depth;
    }
    if ((uint64_t)nv > w->st.max_vertices)
        w->st.max_vertices = (uint64_t)nv;
    v3 lnext = v3_make(0, 0, 0);
    for (int k = nv - 1; k >= 0; --k) {
        vertex_t* v = &vtx[k];
        v->lnext = lnext;
        v3 contrib = v3_div(v3_scale(v3_mul(v->f, lnext), v->c), v->q);
        v3 diffuse = v3_add(v3_make(0, 0, 0), contrib);
        v3 emission = v->emis_param >= 0 ? param_rgb(scene, v->emis_param) : v3_make(0, 0, 0);
        lnext = v3_div(v3_add(emission, diffuse), v->p);
    }
    *L = lnext;
    return nv;
}

static int wants_grad(const drt_scene_desc* sc, int p)
{
    return p >= 0 && (!sc->requires_grad || sc->requires_grad[p]);
}

static void grad_add(double* out, int p, v3 g)
{
    double* acc = &out[p * 3];
    for (int c = 0; c < 3; ++c)
        acc[c] = acc[c] + g.v[c]; /* vector.hpp:187 */
}

int drt_oracle_render(const drt_scene_desc* scene, const drt_camera_desc* cam,
                      const drt_render_params* rp, int rng_mode, uint32_t oracle_flags,
                      const float* adjoint_rgb, double* out_rgb, double* out_param_grad,
                      drt_oracle_stats* stats,
                      drt_oracle_vertex* vertices, uint64_t max_vertices, uint64_t* n_vertices,
                      uint64_t dump_paths)
{
    if (!scene || !cam || !rp || cam->width <= 0 || cam->height <= 0 || rp->spp <= 0)
        return DRT_ERR_INVALID;
    for (int i = 0; i < scene->n_shapes; ++i) {
        const drt_shape_desc* s = &scene->shapes[i];
        if ((s->type != DRT_SHAPE_PLANE && s->type != DRT_SHAPE_SPHERE && s->type != DRT_SHAPE_MESH && s->type != DRT_SHAPE_USER) ||
            s->material >= scene->n_materials || s->emitter >= scene->n_emitters ||
            (s->type == DRT_SHAPE_MESH && (s->mesh < 0 || s->mesh >= scene->n_meshes)))
            return DRT_ERR_INVALID;
        g_scene = scene;
        if (s->type == DRT_SHAPE_USER && !user_kind_is(s, "disc") && !user_kind_is(s, "box"))
            return DRT_ERR_UNSUPPORTED;      /* (a kind this checker has no restatement of) */
    }
    g_scene = scene;
    for (int i = 0; i < scene->n_materials; ++i) {
        const int t = scene->materials[i].type;
        if (t >= DRT_BXDF_USER) {             /* a caller-defined BxDF kind: this checker restates "coslobe" only */
            const int k = t - DRT_BXDF_USER;
            if (scene->n_kinds == 0 || k >= scene->n_bxdf_kinds || !scene->bxdf_kinds || !scene->bxdf_kinds[k].name ||
                strcmp(scene->bxdf_kinds[k].name, "coslobe") != 0)
                return DRT_ERR_UNSUPPORTED;
        }
    }
    const int W = cam->width, H = cam->height, spp = rp->spp;
    const int unbiased = (oracle_flags & DRT_ORACLE_UNBIASED) != 0;
    const int n_shards = rp->n_shards > 1 ? rp->n_shards : 1;
    const int band = rp->band_rows > 0 ? rp->band_rows : 1;
    const int want_grad = out_param_grad != NULL;
    vertex_t* vtx = (vertex_t*)malloc(sizeof(vertex_t) * ORACLE_MAX_VERTICES * 2);
    if (!vtx)
        return DRT_ERR_OOM;
    vertex_t* vtx2 = vtx + ORACLE_MAX_VERTICES;
    walk_ctx w;
    memset(&w, 0, sizeof w);
    w.scene = scene;
    w.rp = rp;
    w.faithful = (oracle_flags & DRT_ORACLE_FAITHFUL_CONTINUATION) != 0 || unbiased;
    w.zero_dir_miss = (oracle_flags & DRT_ORACLE_ZERO_DIR_MISS) != 0 || unbiased;
    w.max_depth = rp->max_depth > 0 ? rp->max_depth : 0; /* 0 = unlimited (reference) */
    w.vertices = vertices;
    w.max_vertices = max_vertices;
    w.status = DRT_OK;
    if (want_grad)
        memset(out_param_grad, 0, sizeof(double) * 3 * (size_t)scene->n_params);
    if (rng_mode == DRT_ORACLE_RNG_LIBC)
        srand(rp->seed);
    w.rng.mode = rng_mode;
    v3 eye = v3_make(cam->eye[0], cam->eye[1], cam->eye[2]);

    for (int y = 0; y < H && w.status == DRT_OK; ++y) {
        if (n_shards > 1 && (y / band) % n_shards != rp->shard)
            continue;
        for (int x = 0; x < W && w.status == DRT_OK; ++x) {
            size_t pix = (size_t)y * W + x;
            v3 pixel = v3_make(0, 0, 0);
            /* gradient image: what param.grad() gains from this pixel's samples alone */
            double before[3] = {0, 0, 0};
            const int gimg = want_grad && g_gimg_out && g_gimg_param >= 0 && g_gimg_param < scene->n_params;
            if (gimg)
                for (int c = 0; c < 3; ++c) {
                    before[c] = out_param_grad[g_gimg_param * 3 + c];
                    out_param_grad[g_gimg_param * 3 + c] = 0.0; /* fresh accumulator, like zeroing grad() */
                }
            for (int i = 0; i < spp; ++i) {
                uint64_t path = (uint64_t)pix * spp + i;
                w.rng.path_key = drt_rng_path_key(rp->seed, path);
                w.rng.draw = 0;
                w.logging = vertices && path < dump_paths;
                w.log_path = (double)path;
                w.log_ord = 0;
                v3 dir = camera_sample(cam, x, y, &w.rng);
                v3 L0;
                int nv = walk(&w, eye, dir, 0, vtx, &L0);
                pixel = v3_add(pixel, v3_div(L0, 1.0)); /* render.cpp:78, pdf = 1 */
                const int loss_l2 = (oracle_flags & DRT_ORACLE_LOSS_L2) != 0 && adjoint_rgb && !unbiased;
                for (int rep = 0; want_grad && rep < (loss_l2 ? 2 : 1); ++rep) {
                    v3 g = v3_make(1, 1, 1); /* render.cpp:80 */
                    if (adjoint_rgb)
                        g = v3_make(adjoint_rgb[pix * 3], adjoint_rgb[pix * 3 + 1], adjoint_rgb[pix * 3 + 2]);
                    if (loss_l2)      /* diff = radiance - target (SubOp), handed to radiance once per operand of diff * diff */
                        g = v3_mul(v3_sub(L0, g), v3_make(1, 1, 1));
                    if (!unbiased) {
                        /* biased: the forward samples are reused; vector.hpp:420-484 from the root */
                        for (int k = 0; k < nv; ++k) {
                            vertex_t* v = &vtx[k];
                            v3 g1 = v3_div(g, v->p);                   /* ScalarDivBackward :479 */
                            if (wants_grad(scene, v->emis_param))
                                grad_add(out_param_grad, v->emis_param, g1);
                            v3 g2 = v3_div(g1, v->q);                  /* ScalarDivBackward */
                            v3 g3 = v3_scale(g2, v->c);                /* ScalarMulBackward :457 */
                            if (wants_grad(scene, v->color_param)) {
                                v3 df = v3_mul(v->lnext, g3);          /* MulBackward :446 */
                                grad_add(out_param_grad, v->color_param,
                                         v->bdiv ? v3_div(df, v->bscalar) : v3_scale(df, v->bscalar));
                            }
                            g = v3_mul(v->f, g3);                      /* MulBackward :447 */
                        }
                    } else if (nv > 0) {
                        /* unbiased: IntegrateBackward (integrate.hpp:11-24) at every vertex draws a
                         * FRESH direction, evaluates forward(sample) -- a new suffix path -- and
                         * back-propagates grad / pdf through it; the recursion continues down the
                         * NEW path, whose first vertex becomes the next `cur` */
                        vertex_t cur = vtx[0];
                        int depth = 0;
                        for (;;) {
                            v3 g1 = v3_div(g, cur.p);                  /* "/ p" of trace() */
                            if (wants_grad(scene, cur.emis_param))     /* AddBackward: emission first */
                                grad_add(out_param_grad, cur.emis_param, g1);
                            if (cur.material < 0) {
                                /* no BxDF: sampler gives a zero direction without drawing, f = 0, and
                                 * forward() still re-traces the zero-length ray (a constant 0) */
                                v3 Lz;
                                (void)walk(&w, cur.point, v3_make(0, 0, 0), depth + 1, vtx2, &Lz);
                                break;
                            }
                            const drt_material_desc* m = &scene->materials[cur.material];
                            double q;
                            v3 dir_out = bxdf_sample(m, cur.normal, cur.dir_in, &w.rng, &q);
                            vertex_eval(scene, &cur, dir_out);
                            v3 Ls;
                            const uint32_t draw_before = (uint32_t)w.rng.draw;
                            int nv2 = walk(&w, v3_add(cur.point, v3_scale(dir_out, 1e-3)), dir_out, depth + 1, vtx2, &Ls);
                            if (getenv("DRT_ORACLE_TRACE_PATH") && (atoll(getenv("DRT_ORACLE_TRACE_PATH")) == -2 || (uint64_t)atoll(getenv("DRT_ORACLE_TRACE_PATH")) == path))
                                fprintf(stderr, "[oracle] path %llu round %d: theta draw %u, suffix base %u, %d suffix vertices, draws after %u, g = %.9g %.9g %.9g, L' = %.9g %.9g %.9g\n",
                                        (unsigned long long)path, depth, draw_before - 2, draw_before, nv2, (unsigned)w.rng.draw, g.v[0], g.v[1], g.v[2], Ls.v[0], Ls.v[1], Ls.v[2]);
                            v3 seed = v3_div(g1, q);                   /* grad / pdf, integrate.hpp:17 */
                            v3 g3 = v3_scale(seed, cur.c);             /* ScalarMulBackward */
                            if (wants_grad(scene, cur.color_param)) {
                                v3 df = v3_mul(Ls, g3);                /* MulBackward: brdf side */
                                grad_add(out_param_grad, cur.color_param,
                                         cur.bdiv ? v3_div(df, cur.bscalar) : v3_scale(df, cur.bscalar));
                            }
                            if (nv2 == 0)
                                break;     /* the suffix is a constant: nothing below */
                            g = v3_mul(cur.f, g3);                     /* MulBackward: radiance side */
                            cur = vtx2[0];
                            ++depth;
                        }
                    }
                }
                w.st.paths++;
            }
            if (gimg)
                for (int c = 0; c < 3; ++c) {
                    double* acc = &out_param_grad[g_gimg_param * 3 + c];
                    g_gimg_out[pix * 3 + c] = *acc / (double)spp;
                    *acc = before[c] + *acc;
                }
            if (out_rgb) {
                v3 mean = v3_div(pixel, (double)spp); /* render.cpp:82 */
                for (int c = 0; c < 3; ++c)
                    out_rgb[pix * 3 + c] = mean.v[c];
            }
        }
    }
    free(vtx);
    w.st.extreme_draws = w.rng.extreme;
    if (stats)
        *stats = w.st;
    if (n_vertices)
        *n_vertices = w.nvlog;
    return w.status;
}

/* sizeof / offsetof of the ABI records as the C compiler lays them out (host-binding tests) */
#include <stddef.h>
int drt_oracle_abi_layout(int which)
{
    switch (which) {
    case 0: return (int)sizeof(drt_shape_desc);
    case 1: return (int)sizeof(drt_material_desc);
    case 2: return (int)sizeof(drt_emitter_desc);
    case 3: return (int)sizeof(drt_scene_desc);
    case 4: return (int)sizeof(drt_camera_desc);
    case 5: return (int)sizeof(drt_render_params);
    case 6: return (int)sizeof(drt_hip_stats);
    case 7: return (int)sizeof(drt_mesh_desc);
    case 10: return (int)offsetof(drt_shape_desc, p);
    case 11: return (int)offsetof(drt_scene_desc, shapes);
    case 12: return (int)offsetof(drt_camera_desc, eye);
    case 13: return (int)offsetof(drt_render_params, absorb);
    case 14: return (int)offsetof(drt_render_params, batch_paths);
    case 15: return (int)offsetof(drt_hip_stats, ms_kernel);
    default: return -1;
    }
}
