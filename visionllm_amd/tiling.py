"""Host-side any-res tile-grid choice that FEEDS the path (it decides how many tiles an image becomes).

Behavioural contract (checked against fixtures produced by running the reference): the grid ``dynamic_preprocess`` picks
and the number of tiles it returns (VisionLLMv2/visionllmv2/mm_utils.py:39-77, helper :23-37).  Pure arithmetic on image
sizes -- the pixel resampling itself stays with PIL / the CLIP image processor in the reference's dataloader
(not accelerated)."""
from functools import lru_cache


@lru_cache(maxsize=None)
def _grids(min_num, max_num):
    """Every (cols, rows) with min_num <= cols * rows <= max_num, smallest tile count first (stable in (cols, rows))."""
    g = {(c, r) for c in range(1, max_num + 1) for r in range(1, max_num + 1) if min_num <= c * r <= max_num}
    return tuple(sorted(g, key=lambda cr: cr[0] * cr[1]))


def find_closest_aspect_ratio(aspect_ratio, target_ratios, width, height, image_size):
    """The candidate grid whose cols / rows is nearest to the image's aspect ratio.  Among equally near candidates (walked
    in the given order) a later one replaces the choice only if the image has more than half the pixels that grid holds."""
    pixels = width * height
    choice, err = (1, 1), float("inf")
    for cols, rows in target_ratios:
        e = abs(aspect_ratio - cols / rows)
        better = e < err
        tie_and_big_enough = e == err and pixels > 0.5 * image_size * image_size * cols * rows
        if better or tie_and_big_enough:
            choice, err = (cols, rows), min(e, err)
    return choice


def dynamic_tile_grid(orig_width, orig_height, min_num=1, max_num=6, image_size=448, use_thumbnail=True):
    """-> (cols, rows, n_tiles): the grid that is cut and the number of [3, image_size, image_size] tiles returned (grid
    tiles + one thumbnail when the grid has more than one tile)."""
    cols, rows = find_closest_aspect_ratio(orig_width / orig_height, _grids(min_num, max_num), orig_width, orig_height,
                                           image_size)
    n = cols * rows
    return cols, rows, n + (1 if use_thumbnail and n != 1 else 0)


def tile_boxes(cols, rows, image_size):
    """Crop boxes (left, upper, right, lower) in the resized image, row-major (the order the tiles are appended in)."""
    boxes = []
    for idx in range(cols * rows):
        y, x = divmod(idx, cols)
        boxes.append((x * image_size, y * image_size, (x + 1) * image_size, (y + 1) * image_size))
    return boxes
