// Output.h — debug printing of rays (all the reference's Output.h holds, inc/Output.h:7-10)
// plus the image writer the reference leaves to its outer application.
#ifndef RTAMD_OUTPUT_H
#define RTAMD_OUTPUT_H

#include <cstdio>
#include <ostream>
#include <string>
#include <vector>

#include "Camera.h"
#include "RTGeometry.h"

namespace rtamd {

inline std::ostream &operator<<(std::ostream &stream, const Ray &ray) {
    return stream << "ray{" << ray.origin << ", " << ray.direction << "}";
}

// Binary PPM (P6) of an RTPixelColor buffer as Camera::render fills it.  Returns false on I/O failure.
inline bool writePPM(const std::string &path, const std::vector<RTPixelColor> &pixels, int width, int height) {
    if (width <= 0 || height <= 0 || pixels.size() < (size_t)width * (size_t)height) return false;
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    for (size_t i = 0; i < (size_t)width * (size_t)height; ++i) {
        unsigned char rgb[3] = {pixels[i].r, pixels[i].g, pixels[i].b};
        if (std::fwrite(rgb, 1, 3, f) != 3) {
            std::fclose(f);
            return false;
        }
    }
    return std::fclose(f) == 0;
}

} // namespace rtamd
#endif
