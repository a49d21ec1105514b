// ORACLE/_ref - TEST INFRASTRUCTURE ONLY. include/core/splat_data.hpp names gs::geometry::BoundingBox in one declaration (crop_by_cropbox, not compiled here).
#pragma once
namespace gs::geometry { class BoundingBox {}; }
