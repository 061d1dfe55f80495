"""Host-side any-res tiling logic that FEEDS the path (it decides how many tiles an image becomes):
``find_closest_aspect_ratio`` and the grid choice of ``dynamic_preprocess``
(VisionLLMv2/visionllmv2/mm_utils.py:23-77).  Pure Python on image sizes -- the pixel resampling itself stays with
PIL / the CLIP image processor in the reference's dataloader (not accelerated)."""


def find_closest_aspect_ratio(aspect_ratio, target_ratios, width, height, image_size):
    best_ratio_diff = float("inf")
    best_ratio = (1, 1)
    area = width * height
    for ratio in target_ratios:
        target_aspect_ratio = ratio[0] / ratio[1]
        ratio_diff = abs(aspect_ratio - target_aspect_ratio)
        if ratio_diff < best_ratio_diff:
            best_ratio_diff = ratio_diff
            best_ratio = ratio
        elif ratio_diff == best_ratio_diff:
            if area > 0.5 * image_size * image_size * ratio[0] * ratio[1]:
                best_ratio = ratio
    return best_ratio


def dynamic_tile_grid(orig_width, orig_height, min_num=1, max_num=6, image_size=448, use_thumbnail=True):
    """-> (cols, rows, n_tiles): the tile grid ``dynamic_preprocess`` cuts and the number of [3,image_size,image_size]
    tiles it returns (grid tiles + the thumbnail when there is more than one tile)."""
    aspect_ratio = orig_width / orig_height
    target_ratios = set((i, j) for n in range(min_num, max_num + 1) for i in range(1, n + 1) for j in range(1, n + 1)
                        if i * j <= max_num and i * j >= min_num)
    target_ratios = sorted(target_ratios, key=lambda x: x[0] * x[1])
    cols, rows = find_closest_aspect_ratio(aspect_ratio, target_ratios, orig_width, orig_height, image_size)
    blocks = cols * rows
    n = blocks + (1 if use_thumbnail and blocks != 1 else 0)
    return cols, rows, n


def tile_boxes(cols, rows, image_size):
    """Crop boxes (left, upper, right, lower) in the resized image, in the order the reference appends the tiles."""
    tw = cols * image_size
    out = []
    for i in range(cols * rows):
        out.append(((i % (tw // image_size)) * image_size, (i // (tw // image_size)) * image_size,
                    ((i % (tw // image_size)) + 1) * image_size, ((i // (tw // image_size)) + 1) * image_size))
    return out
