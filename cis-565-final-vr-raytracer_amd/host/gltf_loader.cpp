// gltf_loader.cpp — placeholder until the glTF reader lands.
#include "scene.hpp"
namespace rth {
bool loadGltfFile(const std::string& filename, GltfScene&, std::string& error)
{
  error = "glTF reader not built yet: " + filename;
  return false;
}
}  // namespace rth
