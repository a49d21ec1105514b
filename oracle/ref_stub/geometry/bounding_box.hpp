// ORACLE/_ref - TEST INFRASTRUCTURE ONLY. The reference's include/geometry/bounding_box.hpp needs the full glm; SplatData::crop_by_cropbox (not compiled here) and
// the crop-box branch of rasterize() (compiled, never entered: the tests pass no box) only need these members to exist.
#pragma once
#include <glm/glm.hpp>
namespace gs::geometry {
    struct EuclideanTransform {
        glm::mat4 toMat4() const { return glm::mat4(); }
    };
    class BoundingBox {
    public:
        glm::vec3 getMinBounds() const { return glm::vec3(); }
        glm::vec3 getMaxBounds() const { return glm::vec3(); }
        const EuclideanTransform& getworld2BBox() const { return t_; }
    private:
        EuclideanTransform t_;
    };
} // namespace gs::geometry
