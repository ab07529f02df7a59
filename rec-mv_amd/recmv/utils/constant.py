"""Name tables the data path needs (utils/constant.py of the reference): which feature lines a capture has
(FL_INFOS, :133-175) and which ATR human-parsing classes make up a garment region (ATR_PARSING, :199-208).  These are data the
reference's directory layout is keyed on, kept verbatim; everything else of that module (colour maps, template lists) belongs
to tools outside the hot path.  INI_FL_SCALE (:236-244): the radial scale every template feature line starts from in the
start-up registration (engineer/core/fl_optimizer.py:141)."""

_UPPER_LOWER = ['neck', 'left_cuff', 'right_cuff', 'upper_bottom', 'left_pant', 'right_pant']

FL_INFOS = {
    'dance': ['short_sleeve_upper'],
    'anran': ['neck', 'left_cuff', 'right_cuff', 'upper_bottom', 'bottom_curve'],
    'xiaolin': ['neck', 'left_cuff', 'right_cuff', 'bottom_curve'],
    'leyang': ['short_sleeve_upper'],
    'tingting': ['short_sleeve_upper'],
    # synthetic captures
    'female_outfit1': ['neck', 'left_cuff', 'right_cuff', 'bottom_curve'],
    'female_outfit3': ['neck', 'bottom_curve'],
    'male_outfit1': list(_UPPER_LOWER),
    'male_outfit2': list(_UPPER_LOWER),
    # large-pose captures
    'anran_run': ['neck', 'left_cuff', 'right_cuff', 'upper_bottom', 'bottom_curve'],
    'anran_tic': ['neck', 'left_cuff', 'right_cuff', 'upper_bottom', 'bottom_curve'],
    'leyang_jump': ['neck', 'left_cuff', 'right_cuff', 'bottom_curve'],
    'leyang_steps': ['neck', 'left_cuff', 'right_cuff', 'bottom_curve'],
    'anran_dance': ['neck', 'left_cuff', 'right_cuff', 'upper_bottom', 'bottom_curve'],
    'lingteng_dance': list(_UPPER_LOWER),
}
# PeopleSnapshot subjects: upper + lower garment, six lines each
for _name in ('female-3-casual', 'female-3-sport', 'female-4-casual', 'female-4-sport', 'female-6-plaza', 'female-7-plaza',
              'male-1-casual', 'male-1-sport', 'male-2-casual', 'male-2-outdoor', 'male-4-casual', 'male-5-outdoor',
              'male-9-plaza'):
    FL_INFOS[_name] = list(_UPPER_LOWER)

# ATR labels: 0 background, 1 hat, 2 hair, 3 sunglasses, 4 upper-clothes, 5 skirt, 6 pants, 7 dress, 8 belt, 9/10 shoes,
# 11 face, 12/13 legs, 14/15 arms, 16 bag, 17 scarf
ATR_PARSING = {
    'upper': [1, 2, 3, 4, 11, 16, 17, 14, 15],                       # with head and hands
    'bottom': [5, 6, 8],
    'upper_bottom': [1, 2, 3, 4, 5, 7, 8, 11, 16, 17, 14, 15, 6],
}

INI_FL_SCALE = {'neck': 1.5, 'right_cuff': 1.5, 'left_cuff': 1.5, 'left_pant': 1.5, 'right_pant': 1.5, 'upper_bottom': 2.,
                'bottom_curve': 2.}
