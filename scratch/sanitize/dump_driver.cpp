// Host-side sanitizer run (AddressSanitizer + UndefinedBehaviorSanitizer on host code only; no GPU is touched):
// builds scenes of every kind through the host layer, flattens them and has the library build, order, enlarge and pack
// its traversal data for them (rtx_traversal_dump runs the same builders rtx_scene_create does).
//   scratch/sanitize/run.sh
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../ray-tracer_amd/host/ray_tracing.hpp"

using namespace ray_tracing;

static int dump_all(const Mesh& root, const char* name, bool has_brackets) {
    SceneBuilder b;
    flatten_scene(root, b);
    const rtx_scene_desc_t d = b.desc();
    const float views[3][3] = {{13, 2, 3}, {-50, 40, -7}, {0, 0, 900}};
    int fails = 0;
    for (uint32_t mode : {(uint32_t)RTX_TRAVERSAL_STRICT, (uint32_t)RTX_TRAVERSAL_FAST, (uint32_t)RTX_TRAVERSAL_FAST_FUSED}) {
        for (int v = 0; v < 4; v++) {
            const float* view = v < 3 ? views[v] : nullptr;
            const float grow = v == 2 ? 1e-3f : 0.0f;
            size_t n = 0;
            int rc = rtx_traversal_dump(&d, mode, view, grow, nullptr, nullptr, 0, &n);
            if (rc == RTX_ERR_UNSUPPORTED && mode == RTX_TRAVERSAL_FAST_FUSED && has_brackets) continue;  // no packed form with transforms / media
            if (rc != RTX_OK) {
                std::printf("%s mode %u view %d: size query failed: %s\n", name, mode, v, rtx_last_error());
                fails++;
                continue;
            }
            const size_t words = mode == RTX_TRAVERSAL_FAST_FUSED ? 4 : 8;
            std::vector<uint32_t> ents(n * words), rank(n);
            rc = rtx_traversal_dump(&d, mode, view, grow, ents.data(), rank.data(), n, &n);
            if (rc != RTX_OK) {
                std::printf("%s mode %u view %d: dump failed: %s\n", name, mode, v, rtx_last_error());
                fails++;
            }
            // a capacity one short must be refused, not overrun
            if (n > 1 && rtx_traversal_dump(&d, mode, view, grow, ents.data(), rank.data(), n - 1, &n) == RTX_OK) {
                std::printf("%s mode %u: short capacity accepted\n", name, mode);
                fails++;
            }
        }
    }
    if (!has_brackets) {  // the wide form (four-child nodes): built for any bracket-free scene when asked for
        for (float grow : {0.0f, 1e-3f}) {
            size_t nn = 0, ns = 0;
            uint32_t levels = 0;
            int rc = rtx_wide_dump(&d, grow, nullptr, 0, nullptr, nullptr, 0, &nn, &ns, &levels);
            if (rc == RTX_ERR_UNSUPPORTED) continue;
            if (rc != RTX_OK) {
                std::printf("%s wide: size query failed: %s\n", name, rtx_last_error());
                fails++;
                continue;
            }
            std::vector<uint32_t> nodes(nn * 16), prims(ns * 4), wrank(ns);
            if (rtx_wide_dump(&d, grow, nodes.data(), nn, prims.data(), wrank.data(), ns, &nn, &ns, &levels) != RTX_OK) {
                std::printf("%s wide: dump failed: %s\n", name, rtx_last_error());
                fails++;
            }
            if (nn > 1 && rtx_wide_dump(&d, grow, nodes.data(), nn - 1, prims.data(), wrank.data(), ns, &nn, &ns, &levels) == RTX_OK) {
                std::printf("%s wide: short capacity accepted\n", name);
                fails++;
            }
        }
    }
    std::printf("%-24s %zu spheres %zu quads %zu tris %zu objects: %s\n", name, b.sph[0].size(), b.quad[0].size(), b.tri[0].size(), b.objects.size(),
                fails ? "FAILED" : "ok");
    return fails;
}

int main() {
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    int fails = 0;
    {  // spheres: moving, stationary, every material
        World w;
        w.push(Sphere::stationary(Vec3(0, -1000, 0), 1000, std::make_shared<Lambertian>(std::make_shared<CheckerTexture>(0.32f, SolidTexture::rgb(.2f, .3f, .1f), SolidTexture::rgb(.9f, .9f, .9f)))));
        for (int i = 0; i < 500; i++) {
            const Vec3 c(22 * U(rng) - 11, 0.2f, 22 * U(rng) - 11);
            const float m = U(rng);
            if (m < 0.8f) w.push(Sphere::moving(c, c + Vec3(0, 0.5f * U(rng), 0), 0.2f, Lambertian::rgb(U(rng), U(rng), U(rng))));
            else if (m < 0.95f) w.push(Sphere::stationary(c, 0.2f, Metal::rgb(U(rng), U(rng), U(rng))->with_roughness(U(rng))));
            else w.push(Sphere::stationary(c, 0.2f, std::make_shared<Dielectric>(1.5f)));
        }
        fails += dump_all(*Bvh::from(w), "spheres", false);
    }
    {  // one primitive, two primitives
        World w;
        w.push(Sphere::stationary(Vec3(0, 0, -1), 0.5f, Lambertian::rgb(.5f, .5f, .5f)));
        fails += dump_all(*Bvh::from(w), "one sphere", false);
        w.push(std::make_shared<Quad>(Vec3(-1, -1, -2), Vec3(2, 0, 0), Vec3(0, 2, 0), Lambertian::rgb(.1f, .2f, .3f)));
        fails += dump_all(*Bvh::from(w), "sphere + quad", false);
    }
    {  // a triangle grid larger than the LDS budget (half-precision boxes) and a small one (float boxes)
        for (int n : {20, 150}) {
            World w;
            auto m = Lambertian::rgb(.6f, .6f, .6f);
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) {
                    const float x = i * 0.1f, z = j * 0.1f, y0 = 0.05f * std::sin(x * 3) * std::cos(z * 2);
                    w.push(std::make_shared<Triangle>(Vec3(x, y0, z), Vec3(x + 0.1f, y0, z), Vec3(x, y0 + 0.01f, z + 0.1f), m));
                    w.push(std::make_shared<Triangle>(Vec3(x + 0.1f, y0, z), Vec3(x + 0.1f, y0, z + 0.1f), Vec3(x, y0 + 0.01f, z + 0.1f), m));
                }
            fails += dump_all(*Bvh::from(w), n == 20 ? "triangles (LDS size)" : "triangles (HBM size)", false);
        }
    }
    {  // cornell-like: quads, cubes under RotateY / Translate, media, nested Bvh, 40 levels of transforms
        World w;
        auto white = Lambertian::rgb(.73f, .73f, .73f);
        w.push(std::make_shared<Quad>(Vec3(555, 0, 0), Vec3(0, 555, 0), Vec3(0, 0, 555), Lambertian::rgb(.12f, .45f, .15f)));
        w.push(std::make_shared<Quad>(Vec3(0, 0, 0), Vec3(0, 555, 0), Vec3(0, 0, 555), Lambertian::rgb(.65f, .05f, .05f)));
        w.push(std::make_shared<Quad>(Vec3(343, 554, 332), Vec3(-130, 0, 0), Vec3(0, 0, -105), DiffuseLight::rgb(15, 15, 15)));
        w.push(std::make_shared<Quad>(Vec3(0, 0, 0), Vec3(555, 0, 0), Vec3(0, 0, 555), white));
        MeshP box1 = std::make_shared<Cube>(Vec3(0, 0, 0), Vec3(165, 330, 165), white);
        box1 = std::make_shared<Translate>(std::make_shared<RotateY>(box1, 15.0f), Vec3(265, 0, 295));
        MeshP box2 = std::make_shared<Cube>(Vec3(0, 0, 0), Vec3(165, 165, 165), white);
        box2 = std::make_shared<Translate>(std::make_shared<RotateY>(box2, -18.0f), Vec3(130, 0, 65));
        w.push(std::make_shared<ConstantMedium>(box1, 0.01f, SolidTexture::rgb(0, 0, 0)));
        w.push(box2);
        World cluster;
        for (int i = 0; i < 200; i++) cluster.push(Sphere::stationary(Vec3(165 * U(rng), 165 * U(rng), 165 * U(rng)), 10, white));
        w.push(std::make_shared<Translate>(std::make_shared<RotateY>(Bvh::from(cluster), 15.0f), Vec3(-100, 270, 395)));
        MeshP deep = Sphere::stationary(Vec3(100, 100, 100), 30, white);
        for (int i = 0; i < 40; i++) deep = i % 2 ? MeshP(std::make_shared<Translate>(deep, Vec3(1, 0, 1))) : MeshP(std::make_shared<RotateY>(deep, 2.0f));
        w.push(deep);
        fails += dump_all(*Bvh::from(w), "instances + media", true);
    }
    {  // instances that can all be placed in the enclosing tree (one and two levels over primitives at rest), one inside a medium
        World w;
        auto white = Lambertian::rgb(.73f, .73f, .73f);
        w.push(std::make_shared<Quad>(Vec3(0, 0, 0), Vec3(555, 0, 0), Vec3(0, 0, 555), white));
        MeshP box1 = std::make_shared<Cube>(Vec3(0, 0, 0), Vec3(165, 330, 165), white);
        box1 = std::make_shared<Translate>(std::make_shared<RotateY>(box1, 15.0f), Vec3(265, 0, 295));
        w.push(std::make_shared<ConstantMedium>(box1, 0.01f, SolidTexture::rgb(0, 0, 0)));
        w.push(std::make_shared<RotateY>(std::make_shared<Cube>(Vec3(0, 0, 0), Vec3(100, 100, 100), white), -18.0f));
        World cluster;
        for (int i = 0; i < 300; i++) cluster.push(Sphere::stationary(Vec3(165 * U(rng), 165 * U(rng), 165 * U(rng)), 10, white));
        w.push(std::make_shared<Translate>(std::make_shared<RotateY>(Bvh::from(cluster), 15.0f), Vec3(-100, 270, 395)));
        fails += dump_all(*Bvh::from(w), "placed instances", true);
    }
    std::printf(fails ? "FAILED (%d)\n" : "all scenes dumped without a sanitizer report\n", fails);
    return fails ? 1 : 0;
}
