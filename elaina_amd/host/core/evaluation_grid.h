// evaluation_grid.h -- probe of the evaluation grid (reference core/evaluation_grid.h:10-41).
// The pixel -> world mapping itself runs on the device (csrc/wost_device.h eval_point);
// the host keeps the data and a reference-equivalent getter for tools.
#pragma once
#include "common.h"

namespace elaina {

template <unsigned int DIM> class EvaluationGrid;

template <> class EvaluationGrid<2> {
public:
    struct ProbeData {
        float scale{1.0f};
        Vector2f pos{0.0f, 0.0f};  // center
        Vector2f up{0.0f, 1.0f};   // up vector
    };

    EvaluationGrid() = default;
    explicit EvaluationGrid(const json &config)
    {
        // {"mData": {"scale": s, "pos": [x,y], "up": [x,y]}} -- all three keys required, like
        // the reference's NLOHMANN_DEFINE_TYPE_INTRUSIVE binding (evaluation_grid.h:22,39)
        mData.scale = json_get_or_throw<float>(config, "mData/scale");
        const auto pos = json_get_or_throw<std::vector<float>>(config, "mData/pos");
        const auto up = json_get_or_throw<std::vector<float>>(config, "mData/up");
        if (pos.size() != 2 || up.size() != 2) throw std::runtime_error("evaluation_grid: pos/up must have 2 entries");
        mData.pos = {pos[0], pos[1]};
        mData.up = {up[0], up[1]};
    }

    Vector2f getEvaluationPoint(Vector2i pixel, Vector2i frameSize) const
    {
        const float ndcx = 2.0f * (float)pixel.x / (float)frameSize.x + -1.0f;
        const float ndcy = 2.0f * (float)pixel.y / (float)frameSize.y + -1.0f;
        const float ux = mData.up.y, uy = -mData.up.x, vx = mData.up.x, vy = mData.up.y;
        return {mData.scale * (ndcx * ux + ndcy * vx) + mData.pos.x, mData.scale * (ndcx * uy + ndcy * vy) + mData.pos.y};
    }

    ProbeData mData;
};

// reference core/evaluation_grid.h:43-70
template <> class EvaluationGrid<3> {
public:
    struct ProbeData {
        float scale{1.0f};
        Vector3f pos{0.0f, 0.0f, 0.0f};
        Vector3f up{0.0f, 0.0f, 1.0f};     // +z as up vector
        Vector3f right{1.0f, 0.0f, 0.0f};  // +x as right vector
    };

    EvaluationGrid() = default;
    explicit EvaluationGrid(const json &config)
    {
        mData.scale = json_get_or_throw<float>(config, "mData/scale");
        const auto pos = json_get_or_throw<std::vector<float>>(config, "mData/pos");
        const auto up = json_get_or_throw<std::vector<float>>(config, "mData/up");
        const auto right = json_get_or_throw<std::vector<float>>(config, "mData/right");
        if (pos.size() != 3 || up.size() != 3 || right.size() != 3) throw std::runtime_error("evaluation_grid: pos/up/right must have 3 entries");
        mData.pos = {pos[0], pos[1], pos[2]};
        mData.up = {up[0], up[1], up[2]};
        mData.right = {right[0], right[1], right[2]};
    }

    Vector3f getEvaluationPoint(Vector2i pixel, Vector2i frameSize) const
    {
        const float ndcx = 2.0f * (float)pixel.x / (float)frameSize.x + -1.0f;
        const float ndcy = 2.0f * (float)pixel.y / (float)frameSize.y + -1.0f;
        return {mData.scale * (ndcx * mData.right.x + ndcy * mData.up.x) + mData.pos.x,
                mData.scale * (ndcx * mData.right.y + ndcy * mData.up.y) + mData.pos.y,
                mData.scale * (ndcx * mData.right.z + ndcy * mData.up.z) + mData.pos.z};
    }

    ProbeData mData;
};

}  // namespace elaina
