"""Pre-computed 2D detections for the Greedy Box Seeker: drop-ins for
pcdet/models/preprocessed_detector.py — `PreprocessedGLIP` (:7-110: one `.pth` of per-image GLIP `BoxList`s + a
COCO-style meta json) and `PreprocessedDetector` (:112-290: one COCO json per camera).

Same constructor arguments, same `__call__(batch_dict) -> (boxes (D,4) xyxy, labels (D,) 1-based, scores (D,),
batch_idx (D,), cam_idx (D,))` CPU tensors with the reference's dtypes, same assertions on token / file-name
alignment.  Differences that do not change results:

  * the reference needs `maskrcnn_benchmark` importable to unpickle the GLIP `BoxList`s; here an unpickler maps the
    classes of missing modules to plain attribute holders, so the same file loads on a machine without it (a list
    of plain dicts {'bbox', 'scores', 'labels'} or (bbox, scores, labels) tuples is accepted too);
  * the per-label Python relabel loop (:83-85) and the per-annotation list appends (:206-216) are replaced by tables
    built once in the constructor: a call is a handful of tensor concatenations.
"""
import json
import pickle
from pathlib import Path

import torch

ALL_CLASS_NAMES = ['car', 'truck', 'construction_vehicle', 'bus', 'trailer',
                   'barrier', 'motorcycle', 'bicycle', 'pedestrian', 'traffic_cone']


class _Holder:
    """Stand-in for a pickled object whose class cannot be imported (maskrcnn_benchmark's BoxList)."""

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):   # (dict, slots)
            self.__dict__.update(state[0] or {})
            self.__dict__.update(state[1])


class _LenientUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return type(name, (_Holder,), {"__module__": module})


class _LenientPickle:
    """pickle_module for torch.load: the standard pickle with the unpickler above."""
    __name__ = "pickle"
    Unpickler = _LenientUnpickler
    load = staticmethod(lambda f, **kw: _LenientUnpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    Pickler = pickle.Pickler
    PickleError = pickle.PickleError
    UnpicklingError = pickle.UnpicklingError


def load_glip_predictions(path):
    """torch.load of a GLIP prediction file that works without maskrcnn_benchmark (legacy and zip formats)."""
    try:
        return torch.load(path, map_location="cpu", weights_only=False)
    except (ImportError, AttributeError, ModuleNotFoundError):
        return torch.load(path, map_location="cpu", pickle_module=_LenientPickle, weights_only=False)


def _boxlist_fields(item):
    """(bbox (n,4) f32, scores (n,), labels (n,) int64) of one image's prediction in any accepted form."""
    if isinstance(item, dict):
        bbox, scores, labels = item["bbox"], item["scores"], item["labels"]
    elif isinstance(item, (tuple, list)) and len(item) == 3:
        bbox, scores, labels = item
    else:   # BoxList-like: .bbox, .extra_fields{'scores','labels'}
        bbox, scores, labels = item.bbox, item.extra_fields["scores"], item.extra_fields["labels"]
    return (torch.as_tensor(bbox).reshape(-1, 4), torch.as_tensor(scores).reshape(-1), torch.as_tensor(labels).reshape(-1))


class PreprocessedGLIP:
    """pcdet/models/preprocessed_detector.py:7-110."""

    def __init__(self, pred_pth='../data/training_pred/nuscenes_glip_train_pred.pth',
                 meta_coco='../data/training_pred/nuscenes_infos_train_mono3d.coco.json', class_names=None):
        self.all_class_names = list(ALL_CLASS_NAMES)
        self.class_names = self.all_class_names if class_names is None else class_names
        self.glip_bbox_file = pred_pth
        self.glip_bboxes = load_glip_predictions(pred_pth)
        self.meta_info_file = meta_coco
        with open(meta_coco, 'r') as f:
            self.meta_info = json.load(f)
        # identity map (:33-34); applied as a table lookup instead of a Python loop per label
        self.map_catid = {(i + 1): (i + 1) for i in range(len(self.all_class_names))}
        self._lut = torch.zeros((len(self.all_class_names) + 1,), dtype=torch.int64)
        for k, v in self.map_catid.items():
            self._lut[k] = v
        self.token_to_id, self.path_to_id = {}, {}
        for img_id, image in enumerate(self.meta_info['images']):
            self.token_to_id[image['token']] = img_id
            self.path_to_id[image['file_name']] = img_id
        self._fields = {}   # img_id -> (boxes, relabelled labels, scores), converted on first use

    def _image(self, img_id):
        hit = self._fields.get(img_id)
        if hit is None:
            bbox, scores, labels = _boxlist_fields(self.glip_bboxes[img_id])
            if labels.numel() and (int(labels.min()) < 1 or int(labels.max()) >= self._lut.numel()):
                raise KeyError(int(labels.max()))      # the reference's dict lookup fails the same way
            hit = (bbox, self._lut[labels.long()].to(labels.dtype), scores)
            self._fields[img_id] = hit
        return hit

    def infer_nusc(self, batch_dict):
        image_paths = batch_dict['image_paths']
        batch_size = batch_dict['batch_size']
        boxes, labels, scores, idx, cam_idx = [], [], [], [], []
        for b in range(batch_size):
            cur_paths = image_paths[b]
            token = batch_dict['metadata'][b]['token']
            for c in range(6):
                path = str(cur_paths[c])
                img_id = self.path_to_id[path]
                meta = self.meta_info['images'][img_id]
                assert token == meta['token'], f"{token} != {meta['token']}"                                   # :72
                assert path == meta['file_name'], f"Batch {path} does not align with GLIP {meta['file_name']}"  # :77
                c_boxes, c_labels, c_scores = self._image(img_id)
                boxes.append(c_boxes)
                labels.append(c_labels)
                scores.append(c_scores)
                idx.extend([b] * len(c_boxes))
                cam_idx.extend([c] * len(c_boxes))
        return (torch.cat(boxes, dim=0), torch.cat(labels, dim=0), torch.cat(scores, dim=0), torch.tensor(idx),
                torch.tensor(cam_idx))

    def __call__(self, batch_dict):
        if 'image_paths' in batch_dict:
            return self.infer_nusc(batch_dict)
        raise TypeError('need kitti / nusc batch dict!')


class PreprocessedDetector:
    """pcdet/models/preprocessed_detector.py:112-290: COCO json predictions (or ground truth) per camera."""

    def __init__(self, cam_jsons=[], class_names=[]):
        assert len(cam_jsons) > 0
        self.cam_infos = []
        self.name_to_anns = {}
        self.categories = None
        self.img_names = set()
        self.class_names = class_names
        self.infer_cam = len(cam_jsons) == 1
        for json_path in cam_jsons:
            with open(json_path, 'r') as f:
                d = json.load(f)
            for img in d['images']:
                if 'name' not in img:
                    img['name'] = Path(img['file_name']).name
            self.cam_infos.append(d)
            assert self.categories is None or self.categories == d['categories'], 'categories differ!'
            self.categories = d['categories']
        cat_ids = set(x['id'] for x in self.categories)
        if self.class_names == [] or self.class_names is None:
            self.class_names = [x['name'] for x in self.categories]
        self.catid_to_classid = {x['id']: (i + 1) for x in self.categories
                                 for i, cls_name in enumerate(self.class_names) if cls_name == x['name']}
        self.wanted_catids = list(self.catid_to_classid.keys())
        if len(self.catid_to_classid) == 0:
            raise ValueError(f"none of the classes {class_names} occurs in the categories {self.categories}")   # (the reference exit()s, :158)
        for view_infos in self.cam_infos:
            img_id_to_name = {img['id']: img['name'] for img in view_infos['images']}
            for img in view_infos['images']:
                self.img_names.add(img['name'])
                self.name_to_anns.setdefault(img['name'], [])
            for ann in view_infos['annotations']:
                if ann['category_id'] not in cat_ids:          # 1-based files (:185-186)
                    ann['category_id'] = ann['category_id'] - 1
                assert ann['category_id'] in cat_ids, f'{ann} not valid'
                self.name_to_anns[img_id_to_name[ann['image_id']]].append(ann)
        first_img_name = list(self.name_to_anns.keys())[0]
        self.incl_ext = '.jpg' in first_img_name or '.png' in first_img_name
        # per image: the wanted annotations as ready-made rows (boxes, class ids, scores)
        self._rows = {}
        for name, anns in self.name_to_anns.items():
            keep = [a for a in anns if a['category_id'] in self.catid_to_classid]
            self._rows[name] = ([a['bbox'] for a in keep], [self.catid_to_classid[a['category_id']] for a in keep],
                                [1.0 if 'score' not in a else a['score'] for a in keep])     # ground truth has no score (:211)

    def _collect(self, names_per_scene, strict=False):
        boxes, labels, scores, idx, cam_idx = [], [], [], [], []
        for b, names in enumerate(names_per_scene):
            for c, name in enumerate(names):
                if name not in self._rows:
                    if strict:
                        raise ValueError(f'frame_id={name} did not exist in preprocessing')
                    continue
                bx, lb, sc = self._rows[name]
                boxes += bx
                labels += lb
                scores += sc
                idx += [b] * len(bx)
                cam_idx += [c] * len(bx)
        return torch.tensor(boxes), torch.tensor(labels), torch.tensor(scores), torch.tensor(idx), torch.tensor(cam_idx)

    def infer_nusc(self, batch_dict):
        names = [[(Path(p).stem if not self.incl_ext else Path(p).name) for p in batch_dict['image_paths'][b]]
                 for b in range(batch_dict['batch_size'])]
        return self._collect(names)

    def infer_kitti(self, batch_dict):
        names = [[(batch_dict['frame_id'][b] + '.png') if self.incl_ext else batch_dict['frame_id'][b]]
                 for b in range(batch_dict['batch_size'])]
        return self._collect(names, strict=True)     # cam_idx is 0 for every box (:263)

    def __call__(self, batch_dict):
        if 'image_paths' in batch_dict:
            return self.infer_nusc(batch_dict)
        elif 'frame_id' in batch_dict:
            return self.infer_kitti(batch_dict)
        raise TypeError('need kitti / nusc batch dict!')
