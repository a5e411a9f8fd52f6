"""CPU restatement of the reference's on-disk formats either side of the hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing in the product path).  Pinned against
goldens produced by the real reference (oracle/make_golden.py formats -> tests/golden/golden_formats.json).

  * preprocess_pifpaf   -- monoloco/network/process.py:155-218 (after json.load of a predictions file)
  * kitti_txt           -- the text save_txts writes, monoloco/eval/generate_kitti.py:202-253
"""
import json
import math

import numpy as np


def preprocess_pifpaf(annotations, im_size=None, enlarge_boxes=True, min_conf=0.):
    """process.py:155-207.  Works on copies (the reference edits the annotation's bbox in place)."""
    boxes, keypoints = [], []
    enlarge = 1 if enlarge_boxes else 2                                   # process.py:163
    for dic in annotations:
        kps_in = dic['keypoints']
        kps = [kps_in[0:][::3], kps_in[1:][::3], kps_in[2:][::3]]           # process.py:210-218
        box = list(dic['bbox'])
        try:                                                              # process.py:168-177
            conf = dic['score']
            delta_h = (box[3]) / (10 * enlarge)
            delta_w = (box[2]) / (5 * enlarge)
            box[2] += box[0]
            box[3] += box[1]
        except KeyError:                                                  # process.py:178-186
            conf = float(np.mean(np.array(kps[2])))
            delta_h = (box[3] - box[1]) / (7 * enlarge)
            delta_w = (box[2] - box[0]) / (3.5 * enlarge)
            assert delta_h > -5 and delta_w > -5, "Bounding box <=0"
        box[0] -= delta_w                                                 # process.py:188-191
        box[1] -= delta_h
        box[2] += delta_w
        box[3] += delta_h
        if im_size is not None:                                           # process.py:194-198
            box[0] = max(0, box[0])
            box[1] = max(0, box[1])
            box[2] = min(box[2], im_size[0])
            box[3] = min(box[3], im_size[1])
        if conf >= min_conf:                                              # process.py:200-204
            box.append(conf)
            boxes.append(box)
            keypoints.append(kps)
    return boxes, keypoints


def read_pifpaf_text(text, im_size=None, enlarge_boxes=True, min_conf=0.):
    return preprocess_pifpaf(json.loads(text), im_size, enlarge_boxes, min_conf)


def kitti_txt(uv_boxes, xyz, bis, epis, alphas=None, rys=None, hwls=None, zzs_geom=None, tt=(0, 0, 0),
              cat=None, conf_scale=0.035):
    """generate_kitti.py:220-253 for one image, returned as a string."""
    lines = []
    for idx, uv_box in enumerate(uv_boxes):
        xx = float(xyz[idx][0]) - tt[0]                                   # :224-226
        yy = float(xyz[idx][1]) - tt[1]
        zz = float(xyz[idx][2]) - tt[2]
        if zzs_geom is not None:                                          # :228-229
            zz = zzs_geom[idx]
        cam_0 = [xx, yy, zz]
        bi = float(bis[idx])
        epi = float(epis[idx])
        if alphas is not None:                                            # :234-238
            alpha, ry = float(alphas[idx]), float(rys[idx])
            hwl = [float(v) for v in hwls[idx]]
        else:                                                             # :239-241
            alpha, ry, hwl = -10., -10., [0, 0, 0]
        conf = conf_scale * (uv_box[-1]) / (bi / math.sqrt(xx ** 2 + yy ** 2 + zz ** 2))  # :242
        output_list = [alpha] + list(uv_box[:-1]) + hwl + cam_0 + [ry, conf, bi, epi]      # :244
        line = ("%s " % 'Pedestrian') if cat[idx] < 0.1 else ("%s " % 'Cyclist')         # :245-249
        line += "%i %i " % (-1, -1)
        for el in output_list:
            line += "%f " % el
        lines.append(line + "\n")
    return "".join(lines)
