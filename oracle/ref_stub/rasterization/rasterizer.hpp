// ORACLE/_ref - TEST INFRASTRUCTURE ONLY. The strategies take a RenderOutput& in post_backward and never look inside it.
#pragma once
namespace gs::training { struct RenderOutput {}; }
