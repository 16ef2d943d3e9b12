// examples/texture_dump.cpp — decodes an image with the C++ host's ImageTexture loader (PNG / PPM / TGA, the
// formats Textured accepts in place of stbi_load, reference main.cpp:147-165) and writes
// "<width> <height>\n" + RGBA8 bytes.  Used by tests/test_host_cpp.py to check the decoders without a GPU.
#include "Raytracer.hpp"
#include <cstdio>
int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: texture_dump <image> <out.rgba>\n"); return 2; }
    ImageTexture t{std::string(argv[1])};
    if (!t.mSet) { std::fprintf(stderr, "could not decode %s\n", argv[1]); return 1; }
    std::FILE* f = std::fopen(argv[2], "wb");
    if (!f) return 1;
    std::fprintf(f, "%d %d\n", t.mWidth, t.mHeight);
    std::fwrite(t.mImage.data(), 1, t.mImage.size(), f);
    std::fclose(f);
    return 0;
}
