/* Sanitizer run of the CPU oracle (test infrastructure): compiles rt_oracle.c into this program with
 * -fsanitize=address,undefined and drives every exported entry point over a small scene that holds
 * every primitive kind, a selected object, every material kind and a light.  Any out-of-bounds access,
 * use of an uninitialised stack slot that ASan can see, misaligned access, signed overflow or invalid
 * shift aborts the run.  Built and run by `make -C oracle asan`. */
#include "rt_oracle.c"

#include <stdio.h>

int main(void) {
    orc_material mats[5];
    memset(mats, 0, sizeof mats);
    mats[0].type = MAT_LAMBERTIAN; mats[0].albedo[0] = 0.8f; mats[0].albedo[1] = 0.8f; mats[0].diffuse[0] = 0.8f;
    mats[1].type = MAT_METAL; mats[1].albedo[0] = 0.8f; mats[1].albedo[1] = 0.6f; mats[1].albedo[2] = 0.2f; mats[1].fuzz = 0.3f;
    mats[2].type = MAT_DIELECTRIC; mats[2].albedo[0] = mats[2].albedo[1] = mats[2].albedo[2] = 1.0f; mats[2].ior = 1.5f;
    mats[3].type = MAT_EMISSIVE; mats[3].emission[0] = mats[3].emission[1] = mats[3].emission[2] = 4.0f;
    mats[4].type = MAT_LAMBERTIAN; mats[4].albedo[2] = 0.7f; mats[4].diffuse[2] = 0.7f;

    orc_sphere sph[5] = {{0, -100.5f, 3, 100}, {0, 0, 3, 0.5f}, {-1, 0, 3, 0.5f}, {1, 0, 3, 0.5f}, {1, 0, 3, -0.4f}};
    uint32_t mat_id[5] = {0, 4, 1, 2, 2};

    float verts[] = {-2, -0.5f, 5, 2, -0.5f, 5, 2, 1.5f, 5, -2, 1.5f, 5, 0.2f, 0.6f, 2.2f, 0.9f, 0.9f, 2.4f, 0.5f, 1.3f, 2.6f};
    orc_prim prims[7];
    memset(prims, 0, sizeof prims);
    prims[0].type = PRIM_PLANE; prims[0].mat = 0; prims[0].p[1] = -0.5f; prims[0].p[4] = 1.0f;
    prims[1].type = PRIM_SPHERE; prims[1].mat = 4; prims[1].p[2] = 3; prims[1].p[3] = 0.5f; prims[1].selected = 1;
    prims[2].type = PRIM_CUBE; prims[2].mat = 1; prims[2].p[0] = -1.3f; prims[2].p[1] = -0.1f; prims[2].p[2] = 3.2f;
    prims[2].p[3] = prims[2].p[4] = prims[2].p[5] = 0.4f; prims[2].selected = 1;
    prims[3].type = PRIM_SPHERE; prims[3].mat = 2; prims[3].p[0] = 1.1f; prims[3].p[2] = 2.8f; prims[3].p[3] = 0.5f;
    prims[4].type = PRIM_POLYGON; prims[4].mat = 0; prims[4].p[2] = -1.0f; prims[4].p[4] = 0.5f; prims[4].p[5] = 5;
    prims[4].first_vertex = 0; prims[4].n_vertices = 4;
    prims[5].type = PRIM_POLYGON; prims[5].mat = 3; prims[5].first_vertex = 4; prims[5].n_vertices = 3; prims[5].selected = 1;
    {   /* normal and centroid of the triangle */
        float ax = verts[15] - verts[12], ay = verts[16] - verts[13], az = verts[17] - verts[14];
        float bx = verts[18] - verts[12], by = verts[19] - verts[13], bz = verts[20] - verts[14];
        float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx, l = sqrtf(nx * nx + ny * ny + nz * nz);
        prims[5].p[0] = nx / l; prims[5].p[1] = ny / l; prims[5].p[2] = nz / l;
        prims[5].p[3] = (verts[12] + verts[15] + verts[18]) / 3; prims[5].p[4] = (verts[13] + verts[16] + verts[19]) / 3;
        prims[5].p[5] = (verts[14] + verts[17] + verts[20]) / 3;
    }
    prims[6].type = PRIM_CUBE; prims[6].mat = 2; prims[6].p[0] = 0.3f; prims[6].p[1] = -0.3f; prims[6].p[2] = 1.8f;
    prims[6].p[3] = 0.15f; prims[6].p[4] = 0.2f; prims[6].p[5] = 0.15f;

    orc_light lights[2];
    memset(lights, 0, sizeof lights);
    lights[0].position[0] = 2; lights[0].position[1] = 3; lights[0].position[2] = 1;
    lights[0].ambient[0] = lights[0].ambient[1] = lights[0].ambient[2] = 0.05f;
    lights[0].diffuse[0] = lights[0].diffuse[1] = 0.6f; lights[0].diffuse[2] = 0.5f; lights[0].view_pow = 16;
    lights[1] = lights[0]; lights[1].position[0] = -3; lights[1].view_pow = 4;

    orc_camera cam;
    const double c[3] = {0, 0, 0}, d[3] = {0, 0, 1};
    orc_update_viewport(c, d, &cam);

    enum { W = 48, H = 27 };
    static uint8_t rgba[W * H * 4];
    static double accum[W * H * 3];
    uint64_t rays = 0;
    /* spheres only: recursion (samplesPerScatter 1 and 2) and the iterative loop, with and without lights */
    rays += orc_render(5, sph, mat_id, 5, mats, 0, NULL, 0, NULL, NULL, &cam, W, H, 4, 1, 10, 1, 0, 0, 0, H, rgba, accum);
    rays += orc_render(5, sph, mat_id, 5, mats, 0, NULL, 0, NULL, NULL, &cam, W, H, 2, 2, 4, 1, 3, 0, 0, H, rgba, NULL);
    rays += orc_render(5, sph, mat_id, 5, mats, 0, NULL, 0, NULL, NULL, &cam, W, H, 4, 1, 50, 1, 0, 1, 5, 9, rgba, accum);
    rays += orc_render(5, sph, mat_id, 5, mats, 2, lights, 0, NULL, NULL, &cam, W, H, 4, 1, 6, 2, 0, 0, 0, H, rgba, accum);
    /* every primitive kind, the halo, lights; an empty scene; depth 0 */
    rays += orc_render(0, NULL, NULL, 5, mats, 1, lights, 7, prims, verts, &cam, W, H, 4, 1, 8, 3, 0, 0, 0, H, rgba, accum);
    rays += orc_render(0, sph, mat_id, 5, mats, 0, NULL, 0, NULL, NULL, &cam, W, H, 1, 1, 5, 3, 0, 0, 0, H, rgba, accum);
    rays += orc_render(5, sph, mat_id, 5, mats, 0, NULL, 0, NULL, NULL, &cam, W, H, 1, 1, 0, 3, 0, 0, 0, H, rgba, accum);

    static float org[W * H * 3], dir[W * H * 3];
    static int32_t ids[W * H];
    static double t[W * H], nrm[W * H * 3], margin[W * H];
    static uint8_t front[W * H], hexp[W * H];
    orc_primary_rays(&cam, W, H, 3, 1, 1, org, dir);
    orc_first_hit(5, sph, 0, NULL, NULL, 0, W * H, org, dir, 0.001f, ids, t, nrm, front, hexp, margin);
    orc_first_hit(0, NULL, 7, prims, verts, 1, W * H, org, dir, 0.001f, ids, t, nrm, front, hexp, NULL);
    int hits = 0, halos = 0;
    for (int i = 0; i < W * H; i++) { hits += ids[i] >= 0; halos += hexp[i]; }

    /* scatter of every material kind on scripted hits */
    enum { N = 8 };
    uint32_t smat[N] = {0, 1, 2, 3, 2, 2, 1, 0};
    double in_dir[N * 3], point[N * 3], normal[N * 3], u[N * 3], att[N * 3], oo[N * 3], od[N * 3];
    uint8_t ffront[N] = {1, 1, 1, 1, 0, 1, 0, 1}, flag[N];
    for (int i = 0; i < N; i++) {
        in_dir[3 * i] = 0.3 * i - 1; in_dir[3 * i + 1] = -1; in_dir[3 * i + 2] = 0.5;
        point[3 * i] = i; point[3 * i + 1] = 0; point[3 * i + 2] = 3;
        normal[3 * i] = 0; normal[3 * i + 1] = 1; normal[3 * i + 2] = 0;
        u[3 * i] = 0.1 * i; u[3 * i + 1] = 0.9 - 0.1 * i; u[3 * i + 2] = 0.05 + 0.12 * i;
    }
    orc_scatter(N, mats, smat, in_dir, point, normal, ffront, u, att, oo, od, flag);

    double a[3], b[3];
    orc_sample_radiance(5, sph, mat_id, 5, mats, &cam, W, H, 600, 2, 1, 12, 5, 0, a);
    orc_sample_radiance(5, sph, mat_id, 5, mats, &cam, W, H, 600, 2, 1, 12, 5, 1, b);
    uint32_t w[4];
    orc_rng4(1, 2, 3, 4, w);
    uint8_t px[4];
    const double rgb[3] = {0.25, -1.0, 7.0};
    orc_convert_color(rgb, px);
    if (a[0] != b[0] || a[1] != b[1] || a[2] != b[2]) {
        fprintf(stderr, "recursive and iterative radiance differ\n");
        return 1;
    }
    printf("oracle sanitizer run ok (%d-byte reals): %llu rays, %d first hits, %d halo hits, pixel %u %u %u, "
           "refract clamps %ld\n", orc_real_bytes(), (unsigned long long)rays, hits, halos, px[0], px[1], px[2],
           orc_refract_clamped());
    return hits > 0 && halos > 0 ? 0 : 1;
}
